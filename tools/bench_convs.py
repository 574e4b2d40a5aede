"""GPU: per-layer forward / backward time of the three subsampling convs at the C3 shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from speech2text_amd import zip_kernels as zk  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
shapes = [((64, 998, 80, 1), 8, (1, 1), 1), ((64, 996, 80, 8), 32, (2, 2), 0), ((64, 497, 39, 32), 128, (1, 2), 0)]
junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)          # evict the caches between runs
for shp, co, stride, pw in shapes:
    x = torch.randn(*shp, device=dev).requires_grad_(shp[-1] != 1)
    w = (torch.randn(co, shp[-1], 3, 3, device=dev) * 0.1).requires_grad_(True)
    b = torch.zeros(co, device=dev).requires_grad_(True)
    tf = tb = 0.0
    n = 6
    for it in range(n + 2):
        junk.zero_()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        y = zk.conv3x3_nhwc(x, w, b, stride, pad_w=pw)
        e[1].record()
        g = torch.ones_like(y)
        torch.cuda.synchronize()
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        y.backward(g)
        e[2].record()
        torch.cuda.synchronize()
        if it >= 2:
            tf += e[0].elapsed_time(e[1])
            tb += e1.elapsed_time(e[2])
    print(f"conv {shp} -> {co} stride {stride}: fwd {tf / n * 1000:7.0f} us  bwd {tb / n * 1000:7.0f} us", flush=True)
