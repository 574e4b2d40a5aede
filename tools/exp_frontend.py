"""Scratch experiment (GPU): Conv2dSubsampling fwd+bwd time, NCHW vs channels_last."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech2text_amd.model.layer.subsampling import Conv2dSubsampling
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = Conv2dSubsampling(80, 192).to(dev).train()
x = torch.randn(64, 998, 80, device=dev)
lens = torch.full((64,), 998, device=dev)

def run(tag, n=5):
    for _ in range(2):
        xx = x.clone().requires_grad_(True)
        y, _ = m(xx, lens); y.sum().backward()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0
    for _ in range(n):
        xx = x.clone().requires_grad_(True)
        e[0].record(); y, _ = m(xx, lens); e[1].record(); y.sum().backward(); e[2].record()
        torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    print(f"{tag:30s} fwd {tf/n:7.2f} ms  bwd {tb/n:7.2f} ms", flush=True)

run("default NCHW")
if os.environ.get("TRY_CL"):
    sys.exit(0)
    fwd0 = Conv2dSubsampling.forward
    def fwd(self, x, x_lens):
        x = x.unsqueeze(1).contiguous(memory_format=torch.channels_last)
        x = self.conv(x); x = self.convnext(x)
        b, c, t, f = x.size()
        x = x.transpose(1, 2).reshape(b, t, c * f)
        x = self.out(x); x = self.out_whiten(x); x = self.out_norm(x); x = self.dropout(x)
        return x, (x_lens - 7) // 2
    Conv2dSubsampling.forward = fwd
    run("channels_last")
