"""GPU: chunked zipconv backward with / without the edge-scale gradient (how much of the weight
kernel's time it is)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from speech2text_amd import zip_kernels as zk
from tools.bench_gemm import timeit
dev = torch.device("cuda")
for T, C, K, chunk in [(495, 192, 31, 32), (495, 192, 31, 495), (248, 256, 31, 16)]:
    B = 64
    u = torch.randn(T, B, 2 * C, device=dev)
    m8 = torch.zeros(B, T, dtype=torch.uint8, device=dev)
    wc = torch.randn(C, (K + 1) // 2, device=dev) * 0.1
    wk = torch.randn(C, K, device=dev) * 0.1
    bk = torch.zeros(C, device=dev)
    sc = torch.randn(2, C, K, device=dev) * 0.1
    dy = torch.randn(T, B, C, device=dev)
    g_all = tuple(torch.zeros_like(t) for t in (wc, bk, wk, bk, sc))
    g_nos = g_all[:4] + (None,)
    t1 = timeit(lambda: zk.zipconv_backward(u, C, m8, chunk, K, wc, wk, bk, sc, dy, g_all))
    t2 = timeit(lambda: zk.zipconv_backward(u, C, m8, chunk, K, wc, wk, bk, sc, dy, g_nos))
    print(f"T={T} C={C} K={K} chunk={chunk}: backward {t1:.1f} us, without d(edge scale) {t2:.1f} us", flush=True)
