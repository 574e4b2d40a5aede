"""Micro-benchmarks of the hand-written kernels at the C3 benchmark shapes (GPU only).
Prints avg launch time (HIP events) and algorithmic GB/s for each entry point."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech2text_amd import zip_kernels as zk, kernels as K, _native as N

dev = torch.device("cuda", 0)


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def report(name, ms, nbytes):
    print(f"{name:46s} {ms*1e3:9.1f} us   {nbytes/ms/1e6:8.1f} GB/s algorithmic ({nbytes/1e6:.1f} MB)", flush=True)


def attn(T, B, H, qd=32, pd=4):
    Dp = H * (2 * qd + pd)
    qkp = torch.randn(T, B, Dp, device=dev) * 0.5
    pos = torch.randn(2 * T - 1, H * pd, device=dev)
    kpm = torch.zeros(B, T, dtype=torch.bool, device=dev)
    W = zk.relpos_attention_weights(qkp, pos, H, qd, pd, None, kpm)
    ms = timeit(lambda: zk.relpos_attention_weights(qkp, pos, H, qd, pd, None, kpm))
    report(f"relpos_attn_fwd T={T} B={B} H={H}", ms, 4.0 * (qkp.numel() + W.numel()))
    dW = torch.randn_like(W)
    qkp.requires_grad_(True); pos.requires_grad_(True)
    def fb():
        W = zk.relpos_attention_weights(qkp, pos, H, qd, pd, None, kpm)
        W.backward(dW)
        qkp.grad = None; pos.grad = None
    ms2 = timeit(fb)
    report(f"relpos_attn fwd+bwd T={T}", ms2, 4.0 * (2 * qkp.numel() + 3 * W.numel()))
    # the training-step mode: dW as factors (two value applies, dv=12) + head-0 term + delta
    qd_, pd_ = qkp.detach(), pos.detach()
    k8 = kpm.to(torch.uint8)
    pairs = [(torch.randn(T, B, H * 12, device=dev), torch.randn(T, B, H * 12, device=dev), None, 12)
             for _ in range(2)]
    dW0 = torch.randn(B, T, T, device=dev)
    delta = torch.randn(H, B, T, device=dev)
    ms3 = timeit(lambda: zk._attn_bwd_call(qd_, pd_, k8, None, H, qd, pd, W.detach(), None, dW0, pairs, delta))
    report(f"relpos_attn_bwd factored T={T}", ms3, 4.0 * (2 * qkp.numel() + 2 * W.numel()))


def conv(T, B, C, Kk):
    class M(torch.nn.Module):
        def __init__(s):
            super().__init__()
            s.kernel_size = Kk
            s.causal_conv = torch.nn.Conv1d(C, C, (Kk + 1) // 2, groups=C)
            s.chunkwise_conv = torch.nn.Conv1d(C, C, Kk, groups=C, padding=Kk // 2)
            s.chunkwise_conv_scale = torch.nn.Parameter(torch.zeros(2, C, Kk))
    m = M().to(dev)
    u = torch.randn(T, B, 2 * C, device=dev, requires_grad=True)
    mask = torch.zeros(B, T, dtype=torch.bool, device=dev)
    ms = timeit(lambda: zk.glu_chunk_causal_dwconv(u.detach(), C, mask, m, -1))
    report(f"zipconv_fwd T={T} B={B} C={C} K={Kk}", ms, 4.0 * (u.numel() + T * B * C))
    dy = torch.randn(T, B, C, device=dev)
    def fb():
        y = zk.glu_chunk_causal_dwconv(u, C, mask, m, -1)
        y.backward(dy)
        u.grad = None
    ms2 = timeit(fb)
    report(f"zipconv fwd+bwd", ms2, 4.0 * (3 * u.numel() + 3 * T * B * C))


def wgrad():
    shapes = [(31680, 192, 192), (31680, 576, 192), (31680, 192, 576), (31680, 512, 192), (31680, 272, 192),
              (15872, 256, 256), (15872, 768, 256), (15872, 256, 768), (15872, 48, 256), (15872, 576, 256),
              (7936, 384, 384), (7936, 1024, 384), (7936, 384, 1024), (7936, 1152, 384),
              (3968, 512, 512), (3968, 1536, 512), (3968, 512, 1536), (32000, 512, 512), (32000, 500, 512)]
    for R, Nf, Mf in shapes:
        g = torch.randn(R, Nf, device=dev); a = torch.randn(R, Mf, device=dev)
        ws = torch.empty(N.lib().s2t_linear_wgrad_workspace_floats(R, Nf, Mf), device=dev)
        dW = torch.empty(Nf, Mf, device=dev); db = torch.empty(Nf, device=dev)
        def ours():
            N.check(N.lib().s2t_linear_wgrad(N.fp(g), g.stride(0), N.fp(a), a.stride(0), R, Nf, Mf, N.fp(dW),
                                             N.fp(db), 0, N.fp(ws), N.stream()), "wgrad")
        def lib():
            g.t().mm(a); g.sum(dim=0)
        m1 = timeit(ours); m2 = timeit(lib); m3 = timeit(lambda: g.t().mm(a))
        fl = 2.0 * R * Nf * Mf
        print(f"wgrad R={R:6d} N={Nf:5d} M={Mf:5d}  ours {m1*1e3:7.1f} us ({fl/m1/1e9:6.1f} TF)   "
              f"mm+sum {m2*1e3:7.1f} us   mm {m3*1e3:7.1f} us ({fl/m3/1e9:6.1f} TF)", flush=True)


if __name__ == "__main__":
    B = 64
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    if only == "wgrad":
        wgrad()
        sys.exit(0)
    if only == "conv":
        conv(496, B, 192, 31)
        conv(248, B, 256, 31)
        conv(124, B, 256, 15)
        conv(62, B, 256, 15)
        sys.exit(0)
    if only == "attn":
        attn(495, B, 4)
        sys.exit(0)
    for T, H in [(495, 4), (248, 4), (124, 4), (62, 8)]:
        attn(T, B, H)
    conv(495, B, 192, 31)
    conv(248, B, 256, 31)
    conv(124, B, 256, 15)
    x = torch.randn(495, B, 576, device=dev)
    ms = timeit(lambda: zk.swoosh_forward(x, True))
    report("swoosh_fwd 495x64x576", ms, 8.0 * x.numel())
    g = torch.randn_like(x)
    ms = timeit(lambda: zk.swoosh_backward(x, g, True))
    report("swoosh_bwd 495x64x576", ms, 12.0 * x.numel())
    ms = timeit(lambda: zk.balancer_backward(x, g, -1.0, 1.0, 0.5, 5.0, 0.04, 2))
    report("balancer_backward 495x64x576", ms, 16.0 * x.numel())
    bias = torch.zeros(192, device=dev); ls = torch.tensor(1.0, device=dev)
    x2 = torch.randn(495, B, 192, device=dev)
    ms = timeit(lambda: zk.bias_norm(x2, bias, ls))
    report("biasnorm_fwd 495x64x192", ms, 8.0 * x2.numel())
    pcm = torch.randn(B, 160000, device=dev) * 0.1
    tab = K.FbankTables(80, device=dev)
    n = torch.full((B,), 160000, dtype=torch.int64, device=dev)
    ms = timeit(lambda: K.fbank_batch(pcm, n, tab))
    report("fbank 64 x 10 s", ms, 4.0 * pcm.numel() + 4.0 * B * 998 * 80)
