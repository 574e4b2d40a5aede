"""Kernels whose tail is one atomic per column and workgroup (s2t_norm_bypass_bwd, s2t_bypass_up_bwd):
time per launch at the C3 shapes, operands rotated over NSET buffer sets (the step never finds them in
the infinity cache).  S2T_BYPASS_UP_BWD16=0 / S2T_NB_BWD16=0 select the scalar forms; the grid caps of the
16-byte forms were environment knobs while they were tuned and are constants now."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech2text_amd import _native as N
dev = torch.device("cuda")
L = N.lib()
NSET = 6


def timeit(fn, n=30):
    for i in range(6):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


B = 64
for T, D in [(495, 192), (248, 256), (124, 256), (62, 256)]:
    R = T * B
    sets = [[torch.randn(R, D, device=dev) for _ in range(3)] + [torch.empty(R, D, device=dev) for _ in range(2)]
            for _ in range(NSET)]
    bias, bsc = torch.zeros(D, device=dev), torch.full((D,), 0.5, device=dev)
    scales = torch.rand(R, device=dev) + 0.5
    acc = torch.zeros(4 * D, device=dev)

    def nb(i):
        x, o, g, dx, d0 = sets[i % NSET]
        N.check(L.s2t_norm_bypass_bwd(N.fp(x), N.fp(bias), N.fp(scales), N.fp(o), N.fp(bsc), N.fp(g), None, B, R, D,
                                      N.fp(dx), N.fp(d0), N.fp(acc), N.fp(acc[2 * D:]), N.fp(acc[3 * D:]), N.stream()),
                "nb")
    us = timeit(nb)
    print(f"norm_bypass_bwd {T}x{B}x{D}: {us:7.1f} us  {20.0 * R * D / us / 1e3:6.0f} GB/s", flush=True)
for T, D, up in [(495, 256, 2), (495, 256, 4), (495, 256, 8), (248, 256, 2)]:
    Ts = (T + up - 1) // up
    sets = [[torch.randn(T, B, D, device=dev), torch.randn(Ts, B, D, device=dev), torch.randn(T, B, D, device=dev),
             torch.empty(T, B, D, device=dev), torch.empty(Ts, B, D, device=dev)] for _ in range(NSET)]
    sc = torch.full((D,), 0.5, device=dev)
    dsc = torch.zeros(D, device=dev)

    def bu(i):
        o, s, g, do, ds = sets[i % NSET]
        N.check(L.s2t_bypass_up_bwd(N.fp(o), N.fp(s), N.fp(sc), N.fp(g), up, T, B, D, N.fp(do), N.fp(ds), N.fp(dsc),
                                    N.stream()), "bu")
    us = timeit(bu)
    nbytes = 4.0 * T * B * D * (3 + 2.0 / up)
    print(f"bypass_up_bwd {T}x{B}x{D} up={up}: {us:7.1f} us  {nbytes / us / 1e3:6.0f} GB/s", flush=True)
# the Balancer's update pass (balancer_apply_fused_kernel) at the step's standalone sites
for R, C, ld in [(31680, 192, 192), (31680, 192, 384), (15872, 256, 256), (15872, 256, 512), (15872, 256, 768),
                 (7936, 256, 256), (3968, 256, 256)]:
    sets = [[torch.randn(R, ld, device=dev), torch.randn(R, ld, device=dev), torch.empty(R, ld, device=dev)]
            for _ in range(NSET)]
    stats = torch.zeros(4096, device=dev)
    stats[:C] = torch.randn(C, device=dev) * R * 0.1
    stats[1024:1024 + C] = (torch.rand(C, device=dev) + 0.5) * R

    def ba(i):
        x, g, o = sets[i % NSET]
        N.check(L.s2t_balancer_apply(N.fp(x), ld, N.fp(g), ld, R, C, -0.4, 0.4,
                                     0.2, 4.0, 0.04, N.fp(o), ld,
                                     N.fp(stats), -1.0, N.stream()), "ba")
    us = timeit(ba)
    print(f"balancer_apply R={R} C={C} ld={ld}: {us:7.1f} us  {12.0 * R * C / us / 1e3:6.0f} GB/s", flush=True)
# SimpleDownsample's backward: every workgroup ends with `ds` atomics on the same words
for T, D, ds in [(495, 256, 2), (495, 256, 4), (495, 256, 8), (248, 256, 2)]:
    dT = (T + ds - 1) // ds
    sets = [[torch.randn(T, B, D, device=dev), torch.randn(dT, B, D, device=dev), torch.empty(T, B, D, device=dev)]
            for _ in range(NSET)]
    w = torch.full((ds,), 1.0 / ds, device=dev)
    dw = torch.zeros(ds, device=dev)

    def dsb(i):
        src, g, dsrc = sets[i % NSET]
        N.check(L.s2t_downsample_bwd(N.fp(src), N.fp(w), N.fp(g), ds, T, B, D, N.fp(dsrc), N.fp(dw), N.stream()), "ds")
    us = timeit(dsb)
    print(f"downsample_bwd {T}x{B}x{D} ds={ds}: {us:7.1f} us  {4.0 * T * B * D * (2 + 1.0 / ds) / us / 1e3:6.0f} GB/s",
          flush=True)
