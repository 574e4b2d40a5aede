import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech2text_amd import _native as N
from tools.bench_gemm import gemm, timeit, dev
for (M, K, Nn) in [(31680, 192, 384), (31680, 192, 576), (15872, 256, 768)]:
    x = torch.randn(M, K, device=dev); g = torch.randn(M, Nn, device=dev)
    dW = torch.zeros(Nn, K, device=dev); db = torch.zeros(Nn, device=dev)
    t1 = timeit(lambda: gemm(2, g, x, dW, Nn, K, M, colsum=db))
    t2 = timeit(lambda: gemm(2, g, x, dW, Nn, K, M))
    dWt = torch.zeros(K, Nn, device=dev)
    t3 = timeit(lambda: gemm(2, x, g, dWt, K, Nn, M))
    print(M, K, Nn, f"with colsum {t1:.1f} us, without {t2:.1f}, transposed-out {t3:.1f}", flush=True)
