"""GPU: A/B of s2t_gemm_x3p variants selected by environment (read once per process) on a few C3
shapes.  usage: S2T_X3P_PRIO=0 python tools/x3p_ab.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech2text_amd import flat
from speech2text_amd import zip_kernels as zk
from tools.bench_gemm import timeit

dev = torch.device("cuda")
SH = [(31680, 512, 192), (31680, 192, 512), (15872, 768, 256), (15872, 256, 768), (15872, 256, 256),
      (7936, 768, 256), (7936, 256, 768), (3968, 768, 256), (3968, 256, 960)]
CF = [int(t) for t in os.environ.get("X3P_CF", "0,222,321,312,411,2022,2021,2012").split(",")]
torch.manual_seed(0)
ws = {(n, k): torch.nn.Parameter(torch.randn(n, k, device=dev) * 0.1) for (_, n, k) in SH}
store = flat.FlatStore(list(ws.values()))
tot = 0.0
for (M, Nn, K) in SH:
    x = torch.randn(M, K, device=dev)
    res = torch.randn(M, Nn, device=dev)
    ts = [timeit(lambda: zk.x3p_matmul(0, x, ws[(Nn, K)], None, res, tile=t), it=20) for t in CF]
    tot += min(ts)
    print(f"{M:6d} {Nn:5d} {K:5d} | " + " ".join(f"{t:6.1f}" for t in ts) + f" | best {CF[ts.index(min(ts))]}", flush=True)
print(f"sum of best {tot:.1f} us   env: " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("S2T_X3P")))
