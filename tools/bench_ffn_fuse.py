"""FFN out-projection with the Swoosh fused into our NT GEMM's operand staging (and its data gradient
with the derivative in the NN GEMM's epilogue) against library GEMM + separate Swoosh pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech2text_amd import zip_kernels as zk
from tools.bench_gemm import timeit, gemm, swoosh, swd
dev = torch.device("cuda")
for (M, F, D) in [(31680, 384, 192), (31680, 512, 192), (31680, 640, 192), (15872, 576, 256), (15872, 768, 256),
                  (15872, 960, 256), (7936, 768, 256), (3968, 768, 256)]:
    h = torch.randn(M, F, device=dev) * 2
    W = torch.randn(D, F, device=dev) * 0.05
    b = torch.randn(D, device=dev)
    res = torch.randn(M, D, device=dev)
    gy = torch.randn(M, D, device=dev)
    y = torch.empty(M, D, device=dev)
    gemm(0, h, W, y, M, D, F, bias=b, resid=res, pro_a=1)
    ref = torch.nn.functional.linear(swoosh(h, 1), W, b) + res
    e1 = ((y - ref).abs().max() / ref.abs().max()).item()
    t_f = timeit(lambda: gemm(0, h, W, y, M, D, F, bias=b, resid=res, pro_a=1))
    def base_f():
        a = zk.swoosh_forward(h, True)
        return zk.lt_matmul(0, a, W, b, res)
    t_b = timeit(base_f)
    dh = torch.empty(M, F, device=dev)
    gemm(1, gy, W, dh, M, F, D, act_src=h, act_kind=1)
    refd = (gy @ W) * swd(h, 1)
    e2 = ((dh - refd).abs().max() / refd.abs().max()).item()
    t_fd = timeit(lambda: gemm(1, gy, W, dh, M, F, D, act_src=h, act_kind=1))
    def base_d():
        d = zk.lt_matmul(1, gy, W)
        return zk.swoosh_backward(h, d, True)
    t_bd = timeit(base_d)
    print(f"M={M} F={F} D={D}: fwd fused {t_f:6.1f} us vs lt+swoosh {t_b:6.1f} | dgrad fused {t_fd:6.1f} us vs lt+swoosh' {t_bd:6.1f} | err {e1:.1e} {e2:.1e}", flush=True)
