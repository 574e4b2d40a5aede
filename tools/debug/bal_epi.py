"""GPU: s2t_gemm_x3p against s2t_gemm_x3p_bal (Balancer folded into the data gradient's epilogue) on the
ConvNeXt and feed-forward shapes, per tile; the statistics pass is timed separately."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from speech2text_amd import flat, zip_kernels as zk, _native as N

dev = torch.device("cuda", 0)
torch.manual_seed(0)
cfg = (-0.05, 0.05, 0.2, 4.0, 0.04)


def t(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (R, Nw, Kw) in ((601920, 128, 384), (31680, 384, 1536), (15872, 512, 2048)):
    w = torch.nn.Parameter(torch.randn(Nw, Kw, device=dev) * 0.05)
    flat.FlatStore([w])
    g = torch.randn(R, Nw, device=dev)
    h = torch.randn(R, Kw, device=dev)
    stats = torch.zeros(4096, device=dev)
    ts = t(lambda: N.lib().s2t_balancer_stats(h.data_ptr(), Kw, R, Kw, stats.data_ptr(), N.stream()))
    print(f"R={R} {Nw}->{Kw}: stats {ts:.0f} us", flush=True)
    for tile in (2022, 2222, 222, 2021, 2012, 312):
        a = t(lambda: zk.x3p_matmul(1, g, w, act_src=h, act_kind="swoosh_l", tile=tile))
        b = t(lambda: zk.x3p_matmul(1, g, w, act_src=h, act_kind="swoosh_l", tile=tile, bal=cfg))
        print(f"   tile {tile}: plain {a:.0f} us   with Balancer (stats + coef + product) {b:.0f} us   product alone ~{b - ts:.0f}", flush=True)
