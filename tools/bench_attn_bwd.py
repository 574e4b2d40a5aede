"""GPU: the factored attention backward alone at a C3 stack shape (for rocprofv3 per-kernel times)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech2text_amd import zip_kernels as zk
dev = torch.device("cuda", 0)
T, B, H, qd, pd = int(os.environ.get("T", 495)), 64, 4, 32, 4
qkp = torch.randn(T, B, H * (2 * qd + pd), device=dev) * 0.5
pos = torch.randn(2 * T - 1, H * pd, device=dev)
kpm = torch.zeros(B, T, dtype=torch.bool, device=dev)
W = zk.relpos_attention_weights(qkp, pos, H, qd, pd, None, kpm).detach()
k8 = kpm.to(torch.uint8)
pairs = [(torch.randn(T, B, H * 12, device=dev), torch.randn(T, B, H * 12, device=dev), None, 12) for _ in range(2)]
dW0 = torch.randn(B, T, T, device=dev)
delta = torch.randn(H, B, T, device=dev)
nopos = os.environ.get("NOPOS") == "1"
now0 = os.environ.get("NOW0") == "1"
for _ in range(12):
    zk._attn_bwd_call(qkp, None if nopos else pos, k8, None, H, qd, pd, W, None, None if now0 else dW0, pairs, delta)
torch.cuda.synchronize()
