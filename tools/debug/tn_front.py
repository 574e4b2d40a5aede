"""GPU: the frontend's weight-gradient products (contraction over 601 920 / 31 680 rows) on the TN forms;
run once per setting of S2T_TN_W / S2T_TN_BLOCKS / S2T_TN_TILE (read once per process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from speech2text_amd import zip_kernels as zk

dev = torch.device("cuda", 0)
torch.manual_seed(0)


def t(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


out = []
for (R, Nf, Mf) in ((601920, 384, 128), (601920, 128, 384), (31680, 192, 2432), (31680, 1536, 384), (31680, 384, 1536)):
    g = torch.randn(R, Nf, device=dev)
    a = torch.randn(R, Mf, device=dev)
    dW = torch.zeros(Nf, Mf, device=dev)
    us = t(lambda: zk.gemm_tn(g, a, dW, None))
    out.append(f"{R}x{Nf}x{Mf}: {us:.0f} us ({2.0 * R * Nf * Mf / us / 1e6:.0f} TF)")
    del g, a
print(os.environ.get("TAGV", "default"), " | ".join(out), flush=True)
