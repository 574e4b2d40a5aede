import os, sys
sys.path.insert(0, "/root/repo")
import torch
from speech2text_amd import flat
from speech2text_amd import zip_kernels as zk
from tools.bench_gemm import timeit
dev = torch.device("cuda")
torch.manual_seed(0)
for (M, Nn, K) in [(15872, 768, 256), (31680, 512, 192), (15872, 256, 768), (7936, 768, 256)]:
    W = torch.nn.Parameter(torch.randn(Nn, K, device=dev) * 0.1)
    b = torch.nn.Parameter(torch.randn(Nn, device=dev))
    store = flat.FlatStore([W, b])
    x = torch.randn(M, K, device=dev)
    for tile in (222, 312, 321):
        t0 = timeit(lambda: zk.x3p_matmul(0, x, W, b, None, tile=tile), it=20)
        cs = zk.GemmColStats(dev)
        t1 = timeit(lambda: zk.x3p_matmul(0, x, W, b, None, tile=tile, colstats=cs), it=20)
        y = zk.x3p_matmul(0, x, W, b, None, tile=tile)
        cs = zk.GemmColStats(dev)
        zk.x3p_matmul(0, x, W, b, None, tile=tile, colstats=cs)
        torch.cuda.synchronize()
        e1 = (cs.buf[:Nn] - y.sum(0)).abs().max().item() / y.sum(0).abs().max().item()
        e2 = (cs.buf[1024:1024 + Nn] - (y * y).sum(0)).abs().max().item() / (y * y).sum(0).abs().max().item()
        print(f"{M}x{Nn}x{K} tile {tile}: plain {t0:6.1f} us, with stats {t1:6.1f} us  (err {e1:.1e} {e2:.1e})", flush=True)
