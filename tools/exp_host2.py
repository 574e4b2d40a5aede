"""Host-side profile of one C3 training step with the backward pass on the calling thread
(autograd multithreading off), GPU queue drained first: where the Python time of a step goes."""
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
torch.autograd.set_multithreading_enabled(False)
for i in range(2):
    trainer.training_step(batch, i)
hs = []
for i in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    trainer.training_step(batch, i)
    hs.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
print("host enqueue (single thread): %.1f ms" % (1e3 * sorted(hs)[2]), flush=True)
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
trainer.training_step(batch, 0)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
rows = []
for (fn, line, name), (cc, nc, tt, ct, callers) in st.stats.items():
    rows.append((tt, ct, nc, "%s:%d(%s)" % (os.path.basename(fn), line, name)))
tot = sum(r[0] for r in rows)
print("profiled step: %.2f ms in %d functions" % (1e3 * tot, len(rows)))
print("%9s %9s %7s  function" % ("self ms", "cum ms", "calls"))
for tt, ct, nc, nm in sorted(rows, reverse=True)[:70]:
    print("%9.3f %9.3f %7d  %s" % (1e3 * tt, 1e3 * ct, nc, nm))
