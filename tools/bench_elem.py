"""Backward passes of the zipformer's small streaming ops at the C3 shapes: time of the backward
kernels only (HIP events around autograd.grad), algorithmic GB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech2text_amd import zip_kernels as zk
dev = torch.device("cuda")


def time_bwd(make, n=20):
    outs = []
    for _ in range(3):
        y, ins = make()
        torch.autograd.grad(y, ins, torch.ones_like(y))
    ys = [make() for _ in range(n)]
    g = torch.randn_like(ys[0][0])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for y, ins in ys:
        torch.autograd.grad(y, ins, g)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def rep(name, us, nbytes):
    print(f"{name:44s} {us:8.1f} us  {nbytes / us / 1e3:7.0f} GB/s ({nbytes / 1e6:.0f} MB)", flush=True)


B = 64
for T, D in [(496, 192), (248, 256), (124, 256)]:
    x = torch.randn(T, B, D, device=dev, requires_grad=True)
    bias = torch.zeros(D, device=dev, requires_grad=True)
    ls = torch.tensor(1.0, device=dev, requires_grad=True)
    rep(f"biasnorm_bwd {T}x{B}x{D}", time_bwd(lambda: (zk.bias_norm(x, bias, ls), (x, bias, ls))), 12.0 * x.numel())
    o = torch.randn(T, B, D, device=dev, requires_grad=True)
    sc = torch.full((D,), 0.5, device=dev, requires_grad=True)
    rep(f"bypass_bwd {T}x{B}x{D}", time_bwd(lambda: (zk.bypass_combine(o, x, sc), (o, x, sc))), 20.0 * x.numel())
for T, D, ds in [(496, 256, 2), (496, 256, 4), (496, 256, 8)]:
    x = torch.randn(T, B, D, device=dev, requires_grad=True)
    w = torch.zeros(ds, device=dev, requires_grad=True)
    rep(f"downsample_bwd {T}x{B}x{D} ds={ds}", time_bwd(lambda: (zk.simple_downsample(x, w, ds), (x, w))),
        4.0 * x.numel() * (2 + 1.0 / ds))
    src = torch.randn((T + ds - 1) // ds, B, D, device=dev, requires_grad=True)
    sc = torch.full((D,), 0.5, device=dev, requires_grad=True)
    rep(f"bypass_up_bwd {T}x{B}x{D} up={ds}", time_bwd(lambda: (zk.bypass_upsampled(x, src, sc, ds), (x, src, sc))),
        4.0 * x.numel() * (3 + 2.0 / ds))
