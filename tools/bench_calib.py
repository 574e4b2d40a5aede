"""GPU: what the library does on a layer shape in bf16 (one product, half the bytes) and fp32 -- the
memory-system floor next to our six-product kernel.  usage: python tools/bench_calib.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.bench_gemm import timeit
dev = torch.device("cuda")
for (M, N, K) in [(15872, 768, 256), (15872, 256, 768), (31680, 512, 192), (7936, 768, 256)]:
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.1
    xb, wb = x.bfloat16(), w.bfloat16()
    o32 = torch.empty(M, N, device=dev)
    ob = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t32 = timeit(lambda: torch.mm(x, w.t(), out=o32))
    tb = timeit(lambda: torch.mm(xb, wb.t(), out=ob))
    tc = timeit(lambda: o32.copy_(torch.empty_like(o32)))
    ta = timeit(lambda: torch.add(o32, 1.0, out=o32))
    print(f"{M}x{N}x{K}: fp32 mm {t32:6.1f} us  bf16 mm (bf16 out) {tb:6.1f} us  |  fp32 C-sized copy {tc:6.1f} us  in-place add {ta:6.1f} us", flush=True)
