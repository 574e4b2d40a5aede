"""The conv module's depthwise kernels alone at the C3 shapes (s2t_zipconv_fwd_act, s2t_zipconv_bwd_data),
operands rotated over NSET buffer sets; algorithmic GB/s = the projection read once, the outputs written once."""
import os, sys
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
for T, D, K in [(495, 192, 31), (248, 256, 31), (124, 256, 15), (62, 256, 15)]:
    R = T * B
    Kh = (K + 1) // 2
    sets = [[torch.randn(T, B, 2 * D, device=dev), torch.empty(T, B, D, device=dev), torch.empty(T, B, D, device=dev),
             torch.randn(T, B, D, device=dev), torch.empty(T, B, 2 * D, device=dev)] for _ in range(NSET)]
    wc, bc = torch.randn(D, Kh, device=dev) * 0.1, torch.zeros(D, device=dev)
    wk, bk = torch.randn(D, K, device=dev) * 0.1, torch.zeros(D, device=dev)
    sc = torch.randn(2, D, K, device=dev) * 0.1
    lens = torch.randint(T * 3 // 4, T + 1, (B,))
    lens[0] = T
    m8 = (torch.arange(T).unsqueeze(0) >= lens.unsqueeze(1)).to(torch.uint8).to(dev)

    def fw(i):
        u, y, a, dy, du = sets[i % NSET]
        N.check(L.s2t_zipconv_fwd_act(N.fp(u), 2 * D, D, m8.data_ptr(), T, B, D, K, T, N.fp(wc), N.fp(bc), N.fp(wk),
                                      N.fp(bk), N.fp(sc), N.fp(y), N.fp(a), 2, N.stream()), "fwd")

    def bw(i):
        u, y, a, dy, du = sets[i % NSET]
        N.check(L.s2t_zipconv_bwd_data(N.fp(u), 2 * D, D, m8.data_ptr(), T, B, D, K, T, N.fp(wc), N.fp(wk), N.fp(bk),
                                       N.fp(sc), N.fp(dy), N.fp(du), N.stream()), "bwd_data")
    us = timeit(fw)
    print(f"zipconv_fwd_act  {T}x{B}x{D} K={K}: {us:7.1f} us  {16.0 * R * D / us / 1e3:6.0f} GB/s", flush=True)
    us = timeit(bw)
    print(f"zipconv_bwd_data {T}x{B}x{D} K={K}: {us:7.1f} us  {20.0 * R * D / us / 1e3:6.0f} GB/s", flush=True)
