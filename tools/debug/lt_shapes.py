"""Per-shape time of the library GEMMs (s2t_linear_lt) inside a C3 step (HIP events around every
call, side stream off) next to the same call repeated back to back in isolation."""
import collections
import os
import random
import sys

os.environ["S2T_WGRAD_STREAM"] = "0"
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from speech2text_amd import _native as N  # noqa: E402
from speech2text_amd import zip_kernels as zk  # noqa: E402
from speech2text_amd.build_task import TaskFactory  # noqa: E402
from speech2text_amd.trainer import Trainer  # noqa: E402

dev = torch.device("cuda:0")
cfg = bench.c3_config(500)
random.seed(1234); np.random.seed(1234); torch.manual_seed(1234)
task = TaskFactory.get("Pruned_Rnnt")(cfg)
tr = Trainer(**cfg["trainer"]).setup(task, dev); task.train()
batch = bench.make_batch(0, 64, 10.0, 50, 500, dev)
for i in range(4):
    tr.training_step(batch, i)
torch.cuda.synchronize()
rec = []
orig = zk.lt_matmul


def wrapped(mode, x2, w2, bias=None, resid2=None, out_shape=None):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    y = orig(mode, x2, w2, bias, resid2, out_shape)
    e1.record()
    rec.append(((mode, x2.shape[0], w2.shape[0], w2.shape[1], bias is not None, resid2 is not None), e0, e1))
    return y


zk.lt_matmul = wrapped
for i in range(3):
    tr.training_step(batch, 10 + i)
torch.cuda.synchronize()
zk.lt_matmul = orig
agg = collections.defaultdict(list)
for key, e0, e1 in rec:
    agg[key].append(e0.elapsed_time(e1) * 1000.0)
rows = []
for key, ts in agg.items():
    mode, M, Nf, Kf, hb, hr = key
    x = torch.randn(M, Kf if mode == 0 else Nf, device=dev)
    w = torch.randn(Nf, Kf, device=dev) * 0.05
    b = torch.randn(Nf, device=dev) if hb else None
    r = torch.randn(M, Nf if mode == 0 else Kf, device=dev) if hr else None
    for _ in range(3):
        orig(mode, x, w, b, r)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        orig(mode, x, w, b, r)
    e1.record()
    torch.cuda.synchronize()
    iso = e0.elapsed_time(e1) * 100.0
    fl = 2.0 * M * Nf * Kf
    rows.append((sum(ts) / 3, key, len(ts) / 3, sum(ts) / len(ts), iso, fl))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"library GEMMs: {tot / 1000:.2f} ms/step in {sum(r[2] for r in rows):.0f} calls")
for t, key, n, avg, iso, fl in rows[:40]:
    print(f"{t / 1000:6.2f} ms  {n:5.1f} calls  in-step {avg:7.1f} us ({fl / avg / 1e6:5.1f} TF/s)  isolated {iso:7.1f} us "
          f"({fl / iso / 1e6:5.1f})  mode={key[0]} M={key[1]} N={key[2]} K={key[3]} bias={int(key[4])} resid={int(key[5])}")
