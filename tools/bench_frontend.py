"""GPU: Conv2dSubsampling (zipformer frontend) forward + backward at the C3 shape, for a
rocprofv3 kernel trace of the frontend alone (x does not require grad, as in training)."""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from speech2text_amd import flat  # noqa: E402
from speech2text_amd.model.layer.subsampling import Conv2dSubsampling  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
random.seed(0)
m = Conv2dSubsampling(80, 192).to(dev).train()
flat.FlatStore(list(m.parameters()))
x = torch.randn(64, 998, 80, device=dev)
lens = torch.full((64,), 998, device=dev)
for i in range(8):
    y, _ = m(x, lens)
    y.sum().backward()
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
n = int(os.environ.get("FE_ITERS", "10"))
for _ in range(n):
    e[0].record()
    y, _ = m(x, lens)
    e[1].record()
    y.sum().backward()
    e[2].record()
    torch.cuda.synchronize()
    tf += e[0].elapsed_time(e[1])
    tb += e[1].elapsed_time(e[2])
print(f"frontend fwd {tf / n:.2f} ms  bwd {tb / n:.2f} ms", flush=True)
