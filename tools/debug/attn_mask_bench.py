"""GPU: relative-position attention weights forward / backward with a chunk mask (the YAML's
chunked training mode) against the unmasked call, at the C3 stack shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from speech2text_amd import zip_kernels as zk
from tools.bench_gemm import timeit
dev = torch.device("cuda")
H, qd, pd = 4, 32, 4
for T, chunk, left in [(495, 32, 4), (248, 16, 4), (124, 8, 4)]:
    B = 64
    qkp = torch.randn(T, B, H * (2 * qd + pd), device=dev, requires_grad=True)
    pos = torch.randn(2 * T - 1, H * pd, device=dev, requires_grad=True)
    c = torch.arange(T, device=dev) // chunk
    am = torch.logical_or(c.unsqueeze(0) > c.unsqueeze(1), c.unsqueeze(0) < c.unsqueeze(1) - left)
    kpm = torch.zeros(B, T, dtype=torch.bool, device=dev)
    for name, m in (("masked  ", am), ("unmasked", None)):
        W = zk.relpos_attention_weights(qkp, pos, H, qd, pd, m, kpm)
        dW = torch.randn_like(W)
        tf = timeit(lambda: zk.relpos_attention_weights(qkp, pos, H, qd, pd, m, kpm))
        tb = timeit(lambda: torch.autograd.grad(W, (qkp, pos), dW, retain_graph=True))
        print(f"T={T} chunk={chunk} {name}: fwd {tf:6.1f} us  bwd {tb:6.1f} us", flush=True)
