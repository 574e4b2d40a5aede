"""s2t_attn_apply at the C3 stack shapes: time and algorithmic GB/s (W read once + v + out)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech2text_amd import _native as N
from tools.bench_gemm import timeit
dev = torch.device("cuda")
for T, H in [(495, 4), (496, 4), (248, 4), (124, 4), (62, 8)]:
    B, dv = 64, 12
    W = torch.rand(H, B, T, T, device=dev)
    v = torch.randn(T, B, H * dv, device=dev)
    out = torch.empty_like(v)
    for tr in (0, 1):
        f = lambda: N.check(N.lib().s2t_attn_apply(N.fp(W), N.fp(v), T, B, H, dv, tr, N.fp(out), N.stream()), "a")
        us = timeit(f)
        nb = 4.0 * (W.numel() + 2 * v.numel())
        print(f"T={T} H={H} transpose={tr}: {us:8.1f} us  {nb / us / 1e3:7.0f} GB/s ({nb / 1e6:.0f} MB)", flush=True)
