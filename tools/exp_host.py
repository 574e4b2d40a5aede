"""Host time of one C3 training step (GPU queue empty at its start) and its cProfile."""
import os, sys, time, random, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from speech2text_amd.task_factory.rnnt_task import PrunedRnntTask
from speech2text_amd.trainer import Trainer

dev = torch.device("cuda", 0)
cfg = bench.c3_config(500)
torch.manual_seed(1234); random.seed(1234)
task = PrunedRnntTask(cfg)
trainer = Trainer(**cfg["trainer"]).setup(task, dev)
task.train()
batch = bench.make_batch(0, 64, 10.0, 50, 500, dev)
for i in range(4):
    trainer.training_step(batch, i)
torch.cuda.synchronize()
hs, ts = [], []
for i in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    trainer.training_step(batch, i)
    hs.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
print("single step from an empty queue: host enqueue %.1f ms, until GPU done %.1f ms" %
      (1e3 * sorted(hs)[len(hs) // 2], 1e3 * sorted(ts)[len(ts) // 2]), flush=True)
# back-to-back
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(10):
    trainer.training_step(batch, i)
th = time.perf_counter() - t0
torch.cuda.synchronize(); t = time.perf_counter() - t0
print("10 steps back to back: %.1f ms/step, host %.1f ms/step" % (100 * t, 100 * th), flush=True)
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
trainer.training_step(batch, 0)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(38)
print(s.getvalue()[:9000])
