"""Host enqueue time vs GPU time of the C3 step (no sync inside the loop)."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from speech2text_amd.build_task import TaskFactory
from speech2text_amd.trainer import Trainer
dev = torch.device("cuda:0")
cfg = bench.c3_config(500)
random.seed(1234); np.random.seed(1234); torch.manual_seed(1234)
task = TaskFactory.get("Pruned_Rnnt")(cfg)
tr = Trainer(**cfg["trainer"]).setup(task, dev); task.train()
batch = bench.make_batch(0, 64, 10.0, 50, 500, dev)
for i in range(4):
    tr.training_step(batch, i)
torch.cuda.synchronize()
t0 = time.perf_counter(); hs = []
for i in range(8):
    a = time.perf_counter(); tr.training_step(batch, 4 + i); hs.append(time.perf_counter() - a)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host per step ms:", [round(1000 * h, 1) for h in hs])
print(f"host total {1000*(t1-t0)/8:.1f} ms/step, wall with sync {1000*(t2-t0)/8:.1f} ms/step")
# split fwd / bwd / opt host time, synchronised per phase (GPU time per phase)
def phase():
    torch.cuda.synchronize(); a = time.perf_counter()
    loss = task.training_step(batch, 0); b = time.perf_counter(); torch.cuda.synchronize(); c = time.perf_counter()
    loss.backward(); d = time.perf_counter(); torch.cuda.synchronize(); e = time.perf_counter()
    tr.optimizer.step(); tr.scheduler.step(); f = time.perf_counter(); torch.cuda.synchronize(); g = time.perf_counter()
    return [1000 * x for x in (b - a, c - a, d - c, e - c, f - e, g - e)]
for _ in range(3):
    r = phase()
print("fwd host %.1f total %.1f | bwd host %.1f total %.1f | opt host %.1f total %.1f" % tuple(r))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU]) as prof:
    tr.training_step(batch, 20)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=35, max_name_column_width=60))
