"""hipBLASLt plan quality at the C2 conformer FFN shapes (M = 248*32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech2text_amd import zip_kernels as zk
from tools.bench_gemm import timeit
dev = torch.device("cuda")
for (M, K, N) in [(7936, 256, 2048), (7936, 2048, 256), (7936, 256, 256), (7936, 256, 768), (7936, 256, 512)]:
    x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev)
    g = torch.randn(M, N, device=dev)
    t0 = timeit(lambda: zk.lt_matmul(0, x, W, b))
    t1 = timeit(lambda: zk.lt_matmul(0, x, W, b, r))
    t2 = timeit(lambda: zk.lt_matmul(1, g, W))
    t3 = timeit(lambda: torch.nn.functional.linear(x, W, b))
    fl = 2.0 * M * K * N
    print(f"M={M} K={K} N={N}: fwd {t0:7.1f} us ({fl/t0/1e6:5.0f} TF)  fwd+resid {t1:7.1f} us  dgrad {t2:7.1f} us ({fl/t2/1e6:5.0f} TF)  torch linear {t3:7.1f} us", flush=True)
