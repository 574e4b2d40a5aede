"""Own NT kernel (bf16x3) and the library at fixed M, N over K: fixed cost per launch vs per-k cost."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech2text_amd import zip_kernels as zk
from tools.bench_gemm import timeit, gemm
dev = torch.device("cuda")
for M, N in [(31680, 512), (15872, 768)]:
    for K in (32, 64, 128, 192, 256, 384, 512, 768):
        x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
        y = torch.empty(M, N, device=dev)
        t_own = timeit(lambda: gemm(0, x, W, y, M, N, K, bias=b))
        t_lt = timeit(lambda: torch.nn.functional.linear(x, W, b))
        print(f"M={M} N={N} K={K:4d}: own {t_own:6.1f} us   lib {t_lt:6.1f} us", flush=True)
