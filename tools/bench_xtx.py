"""s2t_gemm_xtx (Whiten covariance) at the C3 shapes; S2T_TN_BLOCKS picks the slice count."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech2text_amd import _native as N
from tools.bench_gemm import timeit
dev = torch.device("cuda")
for R, C, cg in [(31680, 192, 192), (15872, 256, 256), (7936, 256, 256), (3968, 256, 256), (31680, 128, 32)]:
    x = torch.randn(R, C, device=dev)
    xtx = torch.zeros(C, C, device=dev); cs = torch.zeros(C, device=dev)
    f = lambda: N.check(N.lib().s2t_gemm_xtx(N.fp(x), C, R, C, cg, N.fp(xtx), C, N.fp(cs), N.stream()), "xtx")
    us = timeit(f)
    print(f"R={R} C={C} cg={cg}: {us:7.1f} us  ({2.0*R*C*C/us/1e6:5.1f} TF full-product equivalent)", flush=True)
