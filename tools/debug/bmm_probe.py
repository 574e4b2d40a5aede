"""GPU: the nonlinear attention's three batched products, torch.bmm (rocBLAS) against
s2t_gemm_f32_batched, error against fp64 and time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from speech2text_amd import zip_kernels as zk
from tools.bench_gemm import timeit
dev = torch.device('cuda')
for (B, T, C) in [(64, 495, 144), (64, 248, 192), (64, 124, 192), (64, 62, 192), (64, 248, 144), (8, 100, 36)]:
    W = torch.rand(B, T, T, device=dev); x = torch.randn(B, T, C, device=dev); dz = torch.randn(B, T, C, device=dev)
    cases = [("W@x   ", 1, W, x, lambda: torch.bmm(W, x)), ("W^T@dz", 2, W, dz, lambda: torch.bmm(W.transpose(1, 2), dz)),
             ("dz@x^T", 0, dz, x, lambda: torch.bmm(dz, x.transpose(1, 2)))]
    for name, mode, a, b, ref in cases:
        r64 = {0: lambda: a.double() @ b.double().transpose(1, 2), 1: lambda: a.double() @ b.double(),
               2: lambda: a.double().transpose(1, 2) @ b.double()}[mode]()
        own = zk.batched_matmul(mode, a, b)
        lib = ref()
        e_own = ((own.double() - r64).abs().max() / r64.abs().max()).item()
        e_lib = ((lib.double() - r64).abs().max() / r64.abs().max()).item()
        t_own = timeit(lambda: zk.batched_matmul(mode, a, b)); t_lib = timeit(ref)
        print(f"B={B} T={T} C={C} {name}: own {t_own:6.1f} us (err {e_own:.1e})  bmm {t_lib:6.1f} us (err {e_lib:.1e})", flush=True)
