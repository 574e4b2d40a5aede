"""GPU: the conformer Subsampling conv2 (256 -> 256, 3x3, stride 2, C2 shape) forward / data gradient /
weight gradient: implicit-operand x3p GEMM per tile choice against MIOpen (F.conv2d channels-last)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from speech2text_amd import zip_kernels as zk, planes
from speech2text_amd import _native as N
from tools.bench_gemm import timeit

dev = torch.device("cuda")
torch.manual_seed(0)
B, H, W, C, Cout = 32, 498, 39, 256, 256
x = torch.randn(B, H, W, C, device=dev)
w = torch.randn(Cout, C, 3, 3, device=dev) * 0.02
b = torch.randn(Cout, device=dev)
Ho, Wo = (H - 3) // 2 + 1, (W - 3) // 2 + 1
M = B * Ho * Wo
fl = 2.0 * M * 9 * C * Cout
w2 = w.permute(0, 2, 3, 1).reshape(Cout, 9 * C).contiguous()
pp = planes.adhoc_pieces(w2, 0)
y = torch.empty(M, Cout, device=dev)
amap = zk.RowMap(Ho * Wo, Wo, H * W * C, 2 * W * C, 2 * C, 0)
for tile in (22, 21, 12):
    t = timeit(lambda: N.check(zk._x3p_map(x, amap, 3 * C, [0, W * C, 2 * W * C], pp, Cout, y, Cout, None, 0, M, b, tile), "map"), it=10)
    print(f"forward map tile {tile}: {t:8.1f} us  {fl / t / 1e6:6.1f} TF", flush=True)
xn = x.permute(0, 3, 1, 2)
wcl = w.contiguous(memory_format=torch.channels_last)
t = timeit(lambda: F.conv2d(xn, wcl, b, 2), it=10)
print(f"forward MIOpen      : {t:8.1f} us  {fl / t / 1e6:6.1f} TF")
yr = F.conv2d(xn, wcl, b, 2).permute(0, 2, 3, 1).reshape(M, Cout)
print("fwd max diff", (y - yr).abs().max().item())
g = torch.randn(B, Ho, Wo, Cout, device=dev)
xg = x.clone().requires_grad_(True)
wg = w.clone().requires_grad_(True)
yy = zk.conv3x3_s2_map(xg, wg, None)
t = timeit(lambda: torch.autograd.grad(yy, [xg], g, retain_graph=True), it=10)
print(f"dgrad map (4 classes + pad + weight prep): {t:8.1f} us  {fl / t / 1e6:6.1f} TF")
gy = g.permute(0, 3, 1, 2)
t = timeit(lambda: torch.ops.aten.convolution_backward(gy, xn, wcl, None, [2, 2], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False]), it=10)
print(f"dgrad MIOpen: {t:8.1f} us  {fl / t / 1e6:6.1f} TF")
t = timeit(lambda: zk._conv3x3_wgrad_implicit(x, g.view(-1, Cout), 2, 2, True), it=10)
print(f"wgrad own TN implicit: {t:8.1f} us  {fl / t / 1e6:6.1f} TF")
t = timeit(lambda: torch.ops.aten.convolution_backward(gy, xn, wcl, None, [2, 2], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False]), it=10)
print(f"wgrad MIOpen: {t:8.1f} us  {fl / t / 1e6:6.1f} TF")
# ---- the data gradient's pieces
gp = F.pad(g, (0, 0, 1, 1, 1, 1))
t = timeit(lambda: F.pad(g, (0, 0, 1, 1, 1, 1)), it=10)
print(f"  pad: {t:8.1f} us")
dx = torch.empty(B, H, W, C, device=dev)
P2 = Wo + 2
for ph in (0, 1):
    for pw in (0, 1):
        taps = [(kh, kw) for kh in ((0, 2) if ph == 0 else (1,)) for kw in ((0, 2) if pw == 0 else (1,))]
        t0 = timeit(lambda: torch.cat([w[:, :, kh, kw].t() for kh, kw in taps], dim=1).contiguous(), it=10)
        bc = torch.cat([w[:, :, kh, kw].t() for kh, kw in taps], dim=1).contiguous()
        t1 = timeit(lambda: planes.adhoc_pieces(bc, 0), it=10)
        ppc = planes.adhoc_pieces(bc, 0)
        Hc, Wc = (H - ph + 1) // 2, (W - pw + 1) // 2
        am = zk.RowMap(Hc * Wc, Wc, (Ho + 2) * P2 * Cout, P2 * Cout, Cout, (P2 + 1) * Cout)
        so = [((-1 if kh == 2 else 0) * P2 + (-1 if kw == 2 else 0)) * Cout for kh, kw in taps]
        cm = zk.RowMap(Hc * Wc, Wc, H * W * C, 2 * W * C, 2 * C, (ph * W + pw) * C)
        Mc = B * Hc * Wc
        flc = 2.0 * Mc * len(taps) * Cout * C
        for tile in (22, 21, 12):
            t2 = timeit(lambda: N.check(zk._x3p_map(gp, am, Cout, so, ppc, C, dx, C, cm, dx.numel(), Mc, None, tile), "m"), it=10)
            print(f"  class ({ph},{pw}) taps {len(taps)} tile {tile}: gemm {t2:8.1f} us {flc / t2 / 1e6:6.1f} TF   (weights {t0:.1f} us, pieces {t1:.1f} us)", flush=True)
