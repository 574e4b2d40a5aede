"""GPU: the layer path's streaming kernels at the C3 shapes (B = 64 x 10 s), one line each: time and
algorithmic GB/s.  usage: python tools/bench_side.py [balancer|zipconv|whiten|biasnorm|all]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech2text_amd import _native as N
from speech2text_amd import zip_kernels as zk
from tools.bench_gemm import timeit

dev = torch.device("cuda")
what = sys.argv[1] if len(sys.argv) > 1 else "all"
torch.manual_seed(0)
junk = torch.empty(256 << 20, dtype=torch.uint8, device=dev)


def rep(name, us, nbytes):
    print(f"{name:52s} {us:8.1f} us  {nbytes / us / 1e3:7.0f} GB/s ({nbytes / 1e6:.0f} MB)", flush=True)


def cold(fn, it=10):
    """mean time with the caches flushed before every call (what a kernel sees inside the step)"""
    fn()
    tot = 0.0
    for _ in range(it):
        junk.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / it * 1e3


if what in ("balancer", "all"):
    cfg = (-0.1, 0.1, 0.2, 4.0, 0.04, 1)      # min_mean, max_mean, min_rms, max_rms, grad_scale, channel_dim
    for rows, C in [(31680, 192), (31680, 512), (15872, 768), (15872, 256), (7936, 768), (3968, 768)]:
        x = torch.randn(rows, C, device=dev)
        g = torch.randn(rows, C, device=dev)
        rep(f"balancer_bwd {rows}x{C}", cold(lambda: zk.balancer_backward(x, g, *cfg)), 16.0 * rows * C)
        rep(f"balancer_bwd {rows}x{C} + swoosh'", cold(lambda: zk.balancer_backward(x, g, *cfg, swoosh_l=True)),
            16.0 * rows * C)
    u = torch.randn(15872, 512, device=dev)
    du = torch.randn(15872, 512, device=dev)
    rep("balancer_bwd 15872x256 slice of 512, in place",
        cold(lambda: zk.balancer_backward(u[:, 256:], du[:, 256:], *cfg, inplace=True)), 16.0 * 15872 * 256)

if what in ("zipconv", "all"):
    for T, C, K, chunk in [(495, 192, 31, 0), (248, 256, 31, 0), (124, 256, 15, 0), (62, 256, 15, 0),
                           (495, 192, 31, 32), (248, 256, 31, 16), (124, 256, 15, 8)]:
        B = 64
        chunk = chunk or T                         # chunk < T: the chunk-causal (GEN) kernels
        u = torch.randn(T, B, 2 * C, device=dev)
        m8 = torch.zeros(B, T, dtype=torch.uint8, device=dev)
        wc = torch.randn(C, (K + 1) // 2, device=dev) * 0.1
        bc = torch.zeros(C, device=dev)
        wk = torch.randn(C, K, device=dev) * 0.1
        bk = torch.zeros(C, device=dev)
        sc = torch.randn(2, C, K, device=dev) * 0.1
        y = zk.zipconv_forward(u, C, m8, chunk, K, wc, bc, wk, bk, sc)
        rep(f"zipconv_fwd T={T} C={C} K={K} chunk={chunk}", cold(lambda: zk.zipconv_forward(u, C, m8, chunk, K, wc, bc, wk, bk, sc)),
            4.0 * (u.numel() + y.numel()))
        dy = torch.randn_like(y)
        grads = tuple(torch.zeros_like(t) for t in (wc, bc, wk, bk, sc))
        rep(f"zipconv_bwd T={T} C={C} K={K} chunk={chunk}",
            cold(lambda: zk.zipconv_backward(u, C, m8, chunk, K, wc, wk, bk, sc, dy, grads)),
            4.0 * (2 * u.numel() + 2 * y.numel() + u.numel()))

if what in ("whiten", "all"):
    for rows, C, G in [(31680, 192, 1), (15872, 256, 1), (7936, 256, 1)]:
        x = torch.randn(rows, C, device=dev) * torch.linspace(0.2, 3.0, C, device=dev)
        g = torch.randn(rows, C, device=dev)
        st = zk.WhitenStats(x, G)
        torch.cuda.synchronize()
        rep(f"whiten stats {rows}x{C}", cold(lambda: zk.WhitenStats(x, G).metric()), 4.0 * rows * C)
        rep(f"whiten_bwd {rows}x{C}", cold(lambda: zk.whiten_backward(x, g, st, 1.0, 0.01)), 4.0 * rows * C * 5)

if what in ("biasnorm", "all"):
    import ctypes
    for rows, D in [(31680, 192), (15872, 256), (7936, 256), (3968, 256), (7936, 512)]:
        x = torch.randn(rows, D, device=dev)
        b = torch.randn(D, device=dev) * 0.1
        sc = torch.rand(rows, device=dev) + 0.5
        g = torch.randn_like(x)
        dx = torch.empty_like(x)
        acc = torch.zeros(D + 1, device=dev)

        def call():
            N.check(N.lib().s2t_biasnorm_bwd(N.fp(x), N.fp(b), N.fp(sc), N.fp(g), rows, D, N.fp(dx), N.fp(acc),
                                             ctypes.c_void_p(acc.data_ptr() + 4 * D), N.stream()), "biasnorm_bwd")
        rep(f"biasnorm_bwd {rows}x{D} (cold)", cold(call, it=20), 12.0 * rows * D)
        rep(f"biasnorm_bwd {rows}x{D} (back to back)", timeit(call), 12.0 * rows * D)
