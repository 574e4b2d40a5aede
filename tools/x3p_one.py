"""GPU: a few launches of the bf16x3 GEMM variants on one shape, for rocprofv3 (kernel trace / PMC).
usage: python tools/x3p_one.py M N K [iters]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech2text_amd import _native as N
from speech2text_amd import flat
from speech2text_amd import zip_kernels as zk

M, Nn, K = (int(a) for a in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda")
torch.manual_seed(0)
W = torch.nn.Parameter(torch.randn(Nn, K, device=dev) * 0.1)
b = torch.nn.Parameter(torch.randn(Nn, device=dev))
store = flat.FlatStore([W, b])
x = torch.randn(M, K, device=dev)
out = torch.empty(M, Nn, device=dev)
L = N.lib()
for t in (22, 21, 12, 11):
    for _ in range(iters):
        zk.x3p_matmul(0, x, W, b, None, tile=t)
for t in (22, 23, 21):
    for _ in range(iters):
        N.check(L.s2t_gemm_f32_tiled(0, N.fp(x), K, N.fp(W.detach()), K, N.fp(out), Nn, M, Nn, K, N.fp(b.detach()),
                                     None, 0, t, N.stream()), "tiled")
os.environ["S2T_LT_OWN"] = "0"
for _ in range(iters):
    zk._lt_matmul_lib(0, x, W.detach(), b.detach(), None)
torch.cuda.synchronize()
print("done")
