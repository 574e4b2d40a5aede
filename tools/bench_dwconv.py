"""GPU: timing of the channel-last depthwise 7x7 conv (frontend ConvNeXt) at the C3 shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from speech2text_amd import _native as N  # noqa: E402
from tools.bench_gemm import dev, timeit  # noqa: E402

L = N.lib()
for (Nn, H, W, C) in [(64, 495, 19, 128), (8, 495, 19, 128), (64, 495, 19, 64), (64, 124, 19, 128)]:
    x = torch.randn(Nn, H, W, C, device=dev)
    w = torch.randn(C, 7, 7, device=dev)
    b = torch.randn(C, device=dev)
    y = torch.empty_like(x)
    t = timeit(lambda: L.s2t_dwconv2d_nhwc_fwd(N.fp(x), N.fp(w), N.fp(b), Nn, H, W, C, 7, 7, 0, N.fp(y), N.stream()))
    tc = timeit(lambda: y.copy_(x))
    tm = timeit(lambda: torch.mul(x, 2.0, out=y))
    mb = x.numel() * 4 / 1e6
    print(f"N={Nn} H={H} W={W} C={C}: dwconv {t:7.1f} us  ({2*mb/t/1e0:6.1f} MB/us... {2 * mb / t * 1e-3:.2f} TB/s, "
          f"{x.numel() * 98 / t / 1e6:.1f} TFLOP/s) | copy {tc:6.1f} us | mul {tm:6.1f} us", flush=True)
    dy = torch.randn_like(x)
    ws = torch.empty(L.s2t_dwconv2d_wgrad_workspace_floats(Nn, H, C, 7, 7), device=dev)
    dw = torch.empty(C, 7, 7, device=dev)
    db = torch.empty(C, device=dev)
    tw = timeit(lambda: L.s2t_dwconv2d_nhwc_wgrad(N.fp(x), N.fp(dy), Nn, H, W, C, 7, 7, N.fp(ws), N.fp(dw), N.fp(db), N.stream()))
    ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2), (C, 1, 7, 7), dy.permute(0, 3, 1, 2), padding=3, groups=C)
    err = ((dw - ref[:, 0]).abs().max() / ref.abs().max()).item()
    print(f"    wgrad {tw:7.1f} us  ({2 * mb / tw * 1e-3:.2f} TB/s)  err {err:.1e}  db err {((db - dy.sum((0, 1, 2))).abs().max() / dy.sum((0, 1, 2)).abs().max()).item():.1e}", flush=True)
