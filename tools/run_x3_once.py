"""GPU: 20 launches of s2t_gemm_x3_nt on one shape (for rocprofv3 --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.bench_x3 import planes_of, x3, dev
M, K, Nn = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (15872, 256, 768)))
x = torch.randn(M, K, device=dev); W = torch.randn(Nn, K, device=dev) * 0.1; b = torch.randn(Nn, device=dev)
pl = planes_of(W); y = torch.empty(M, Nn, device=dev)
for _ in range(20):
    x3(x, pl, Nn, K, y, b)
torch.cuda.synchronize()
