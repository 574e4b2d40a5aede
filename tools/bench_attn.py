"""GPU: time s2t_relpos_attn_fwd at the C3 stack shapes (B=64) and report algorithmic GB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech2text_amd import zip_kernels as zk
dev = torch.device("cuda")
for (T, H) in [(495, 4), (248, 4), (124, 4), (62, 8)]:
    B, qd, pd = 64, 32, 4
    qkp = torch.randn(T, B, H * (2 * qd + pd), device=dev)
    pos = torch.randn(2 * T - 1, H * pd, device=dev)
    kpm = torch.zeros(B, T, dtype=torch.bool, device=dev)
    f = lambda: zk.relpos_attention_weights(qkp, pos, H, qd, pd, None, kpm)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1000
    nbytes = 4.0 * (qkp.numel() + H * B * T * T)
    print(f"T={T} H={H}: {us:.1f} us, {nbytes/1e6:.1f} MB algorithmic -> {nbytes/us/1e3:.0f} GB/s ({nbytes/us/1e3/8000:.3f} of HBM)", flush=True)
