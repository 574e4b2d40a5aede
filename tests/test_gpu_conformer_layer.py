"""GPU: the conformer block's HIP kernels (LayerNorm, SiLU, BatchNorm+SiLU, flash-style MHSA) against
fp64 torch-CPU restatements, and the one-node layer executor (conf_layer.py) against the oracle
conformer (oracle/conformer.py; torchaudio block structure, parity unpinned) incl. every
parameter gradient, up to the C2 dims (d=256, 4 heads, ffn 2048, k=31, 12 layers)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import conformer as OC

pytestmark = pytest.mark.gpu


def _close(a, b, tol, what=""):
    a = a.detach().double().cpu().numpy()
    b = b.detach().double().cpu().numpy()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b).max()
    ref = np.abs(b).max() + 1e-30
    assert err <= tol * ref + 1e-7, f"{what}: max err {err:.3e} vs max |ref| {ref:.3e}"


@pytest.mark.parametrize("rows,C", [(37, 256), (1001, 144), (64, 1024), (5, 64)])
def test_layernorm_fwd_bwd(dev, rows, C):
    from speech2text_amd import conf_kernels as ck
    g = torch.Generator().manual_seed(rows + C)
    x = torch.randn(rows, C, generator=g) * 2 + 0.5
    y = torch.randn(rows, C, generator=g)
    w = torch.randn(C, generator=g)
    b = torch.randn(C, generator=g)
    dy = torch.randn(rows, C, generator=g)
    res = torch.randn(rows, C, generator=g)
    for add in (False, True):
        xd = x.double().requires_grad_(True)
        yd = y.double().requires_grad_(True)
        wd, bd = w.double().requires_grad_(True), b.double().requires_grad_(True)
        xin = xd + 0.5 * yd if add else xd
        ref = F.layer_norm(xin, (C,), wd, bd, 1e-5)
        (ref * dy.double()).sum().backward()
        xsum, out, stats = ck.ln_fwd(x.to(dev), y.to(dev) if add else None, 0.5, w.to(dev),
                                     b.to(dev), 1e-5)
        _close(out, ref, 2e-6, "ln out")
        if add:
            _close(xsum, xin, 1e-6, "ln sum")
        dgam = torch.zeros(C, device=dev)
        dbet = torch.zeros(C, device=dev)
        xs = xsum if add else x.to(dev)
        pend = []
        dx = ck.ln_bwd(xs, stats, w.to(dev), dy.to(dev), res.to(dev), pend)
        ck.ln_param_grad([pend[0] + (dgam, dbet)], C)
        _close(dx, xd.grad + res.double(), 5e-6, "ln dx")
        _close(dgam, wd.grad, 2e-5, "ln dgamma")
        _close(dbet, bd.grad, 2e-5, "ln dbeta")


def test_silu_fwd_bwd(dev):
    from speech2text_amd import conf_kernels as ck
    g = torch.Generator().manual_seed(3)
    h = torch.randn(77, 2048, generator=g) * 3
    da = torch.randn(77, 2048, generator=g)
    hd = h.double().requires_grad_(True)
    ref = F.silu(hd)
    (ref * da.double()).sum().backward()
    a = ck.silu_fwd(h.to(dev))
    _close(a, ref, 2e-6, "silu")
    dh = ck.silu_bwd(h.to(dev), da.to(dev), 0.5, inplace=False)
    _close(dh, 0.5 * hd.grad, 5e-6, "silu bwd")


@pytest.mark.parametrize("rows,C", [(7936, 256), (333, 64), (50, 192)])
def test_batchnorm_silu_fwd_bwd(dev, rows, C):
    from speech2text_amd import conf_kernels as ck
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, C, generator=g) * 1.7 + torch.randn(C, generator=g)
    ds = torch.randn(rows, C, generator=g)
    bn = torch.nn.BatchNorm1d(C)
    with torch.no_grad():
        bn.weight.copy_(torch.randn(C, generator=g))
        bn.bias.copy_(torch.randn(C, generator=g))
        bn.running_mean.copy_(torch.randn(C, generator=g))
        bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    ref_bn = torch.nn.BatchNorm1d(C).double()
    ref_bn.load_state_dict({k: v.double() if v.dtype.is_floating_point else v
                            for k, v in bn.state_dict().items()})
    ref_bn.train()
    xd = x.double().requires_grad_(True)
    ref = F.silu(ref_bn(xd))
    (ref * ds.double()).sum().backward()
    bn = bn.to(dev).train()
    y, mean, rstd = ck.bn_silu_fwd(x.to(dev), bn)
    _close(y, ref, 5e-6, "bn y")
    _close(bn.running_mean, ref_bn.running_mean, 1e-5, "running_mean")
    _close(bn.running_var, ref_bn.running_var, 1e-5, "running_var")
    assert int(bn.num_batches_tracked) == 1
    dgam = torch.zeros(C, device=dev)
    dbet = torch.zeros(C, device=dev)
    dx = ck.bn_silu_bwd(x.to(dev), ds.to(dev), mean, rstd, bn.weight, bn.bias, dgam, dbet)
    _close(dx, xd.grad, 2e-5, "bn dx")
    _close(dgam, ref_bn.weight.grad, 2e-5, "bn dgamma")
    _close(dbet, ref_bn.bias.grad, 2e-5, "bn dbeta")
    # evaluation mode: running statistics
    bn.eval()
    ref_bn.eval()
    _close(ck.batchnorm_silu(x.to(dev), bn), F.silu(ref_bn(x.double())), 5e-6, "bn eval")


def _attn_ref(qkv, lens, H, mask=None):
    """fp64 restatement of nn.MultiheadAttention's core on (T,B,3D); mask (B,H,T,T) of kept
    probabilities scaled by 1/(1-p) (dropout), or None."""
    T, B, D3 = qkv.shape
    D = D3 // 3
    dh = D // H
    q, k, v = (qkv[..., i * D:(i + 1) * D].reshape(T, B, H, dh).permute(1, 2, 0, 3) for i in range(3))
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh)
    if lens is not None:
        kpm = torch.arange(T).view(1, 1, 1, T) >= lens.view(B, 1, 1, 1)
        s = s.masked_fill(kpm, float("-inf"))
    p = s.softmax(-1)
    if mask is not None:
        p = p * mask
    return torch.matmul(p, v).permute(2, 0, 1, 3).reshape(T, B, D)


@pytest.mark.parametrize("T,B,H,dh,ragged", [(248, 4, 4, 64, True), (70, 3, 2, 32, True),
                                             (33, 2, 4, 16, False), (748, 2, 4, 64, True),
                                             (129, 1, 1, 64, True)])
def test_mhsa_fwd_bwd_vs_fp64(dev, T, B, H, dh, ragged):
    from speech2text_amd import conf_kernels as ck
    g = torch.Generator().manual_seed(T * 7 + B)
    D = H * dh
    qkv = torch.randn(T, B, 3 * D, generator=g)
    qkv[..., :D] *= 1.5                                   # peaky rows exercise the running max
    do = torch.randn(T, B, D, generator=g)
    lens = None
    if ragged:
        lens = torch.tensor([T, max(1, T // 2 + 3), 1, max(1, T - 31)][:B])
    qd = qkv.double().requires_grad_(True)
    ref = _attn_ref(qd, lens, H)
    (ref * do.double()).sum().backward()
    q2 = qkv.to(dev).view(T * B, 3 * D).contiguous()
    ld = None if lens is None else lens.to(dev)
    o, lse = ck.mhsa_fwd(q2, ld, T, B, H)
    _close(o.view(T, B, D), ref, 1e-5, "attention out")
    dqkv = ck.mhsa_bwd(q2, ld, T, B, H, o, do.to(dev).view(T * B, D).contiguous(), lse)
    _close(dqkv.view(T, B, 3 * D), qd.grad, 2e-5, "attention dqkv")
    # through the autograd wrapper (the module path)
    qg = qkv.to(dev).requires_grad_(True)
    og = ck.mhsa(qg, ld, H)
    (og * do.to(dev)).sum().backward()
    _close(qg.grad, qd.grad, 2e-5, "attention dqkv (autograd)")


def test_mhsa_dropout_mask_is_consistent(dev):
    """Attention-probability dropout: with V = one-hot rows the output IS the dropped probability
    matrix, so the mask can be read off; forward and backward must then agree with an fp64
    restatement that uses that very mask (the backward regenerates it from the seed)."""
    from speech2text_amd import conf_kernels as ck
    T, B, H, dh, p, seed = 64, 2, 2, 64, 0.25, 123456789
    D = H * dh
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(T, B, 3 * D, generator=g)
    probe = qkv.clone()
    probe[..., 2 * D:] = torch.eye(T).view(T, 1, 1, dh).expand(T, B, H, dh).reshape(T, B, D)
    q2 = probe.to(dev).view(T * B, 3 * D).contiguous()
    o0, _ = ck.mhsa_fwd(q2, None, T, B, H)
    o1, _ = ck.mhsa_fwd(q2, None, T, B, H, p, seed)
    P0 = o0.view(T, B, H, T).permute(1, 2, 0, 3).cpu().double()          # (B,H,q,k)
    P1 = o1.view(T, B, H, T).permute(1, 2, 0, 3).cpu().double()
    kept = P1 > 0
    np.testing.assert_allclose(P1[kept].numpy(), (P0[kept] / (1 - p)).numpy(), rtol=2e-5)
    rate = kept.double().mean().item()
    assert abs(rate - (1 - p)) < 0.03, rate
    mask = kept.double() / (1 - p)
    # random values, same seed -> same mask
    do = torch.randn(T, B, D, generator=g)
    qd = qkv.double().requires_grad_(True)
    ref = _attn_ref(qd, None, H, mask)
    (ref * do.double()).sum().backward()
    q3 = qkv.to(dev).view(T * B, 3 * D).contiguous()
    o, lse = ck.mhsa_fwd(q3, None, T, B, H, p, seed)
    _close(o.view(T, B, D), ref, 1e-5, "dropout attention out")
    dqkv = ck.mhsa_bwd(q3, None, T, B, H, o, do.to(dev).view(T * B, D).contiguous(), lse, p, seed)
    _close(dqkv.view(T, B, 3 * D), qd.grad, 3e-5, "dropout attention dqkv")
    # another seed gives another mask
    o2, _ = ck.mhsa_fwd(q2, None, T, B, H, p, seed + 1)
    assert (o2 > 0).ne(o1 > 0).any()


def _stack_vs_oracle(dev, cfg, B, T, lens, tol_out, tol_grad, seed=0, monkeypatch=None):
    from speech2text_amd import conf_kernels as ck
    from speech2text_amd import conf_layer, flat
    from speech2text_amd.model.encoder.conformer import Conformer, ConformerConfig
    torch.manual_seed(seed)
    m = Conformer(ConformerConfig(**cfg)).to(dev)
    with torch.no_grad():                                  # non-trivial norms / biases
        for n, p in m.named_parameters():
            if n.endswith("bias") or "norm" in n or n.endswith("sequential.3.weight"):
                p.add_(0.1 * torch.randn_like(p))
    flat.get_store([p for p in m.parameters()])
    pnames = {n for n, _ in m.named_parameters()}
    sd = {k: v.detach().cpu().clone().requires_grad_(k in pnames) for k, v in m.state_dict().items()}
    x = torch.randn(B, T, 80)
    m.train()
    n0 = conf_layer.CALLS[0]
    seeds = []
    if monkeypatch is not None:            # record the seeds of the dropout sites, in draw order
        real = ck.draw_seed
        monkeypatch.setattr(ck, "draw_seed", lambda: seeds.append(real()) or seeds[-1])
    xg = x.to(dev).requires_grad_(True)
    y, l = m(xg, lens.to(dev))
    assert conf_layer.CALLS[0] - n0 == cfg["num_layers"], "the layer executor did not run"
    if cfg["dropout"] > 0:
        assert len(seeds) == 7 * cfg["num_layers"], len(seeds)
    xc = x.clone().requires_grad_(True)
    yo, lo = OC.conformer_forward(sd, xc, lens, cfg["num_layers"], cfg["num_heads"], training=True,
                                  dropout=cfg["dropout"], seeds=seeds)
    assert torch.equal(l.cpu(), lo)
    _close(y, yo, tol_out, "encoder out")
    w = torch.randn_like(yo)
    (yo * w).sum().backward()
    (y * w.to(dev)).sum().backward()
    torch.cuda.synchronize()
    _close(xg.grad, xc.grad, tol_grad, "d input")
    for n, p in m.named_parameters():
        ref = sd[n].grad
        got = p.grad.detach().cpu()
        err = (got - ref).abs().max().item()
        if n.endswith("conv_module.sequential.2.bias"):
            # the depthwise conv's bias sits in front of BatchNorm: its true gradient is exactly
            # zero and both sides hold rounding noise of the (large) per-channel sums
            assert err <= 1e-4 * p.shape[0] ** 0.5 * max(1.0, float(w.abs().max())), (n, err)
            continue
        assert err <= tol_grad * ref.abs().max().item() + 3e-4, (n, err, ref.abs().max().item())
    # running statistics were updated like torch's BatchNorm1d does
    for k, v in m.state_dict().items():
        if k.endswith("num_batches_tracked"):
            assert int(v) == 1, k


def test_layer_executor_vs_oracle_small(dev):
    cfg = dict(input_dim=64, num_heads=4, ffn_dim=128, num_layers=2, depthwise_conv_kernel_size=15,
               dropout=0.0, output_dim=40)
    _stack_vs_oracle(dev, cfg, 3, 203, torch.tensor([203, 150, 99]), 2e-4, 1e-2)


def test_layer_executor_with_dropout_vs_oracle(dev, monkeypatch):
    """The YAMLs' dropout 0.1 (config/training/conformer_ctc.yaml:63): all seven nn.Dropout sites of
    a block run inside the executor as hashed masks; output and every gradient against the oracle
    applying the same masks (oracle.conformer.keep_scale restates the hash)."""
    cfg = dict(input_dim=64, num_heads=4, ffn_dim=128, num_layers=2, depthwise_conv_kernel_size=15,
               dropout=0.1, output_dim=40)
    _stack_vs_oracle(dev, cfg, 3, 203, torch.tensor([203, 150, 99]), 3e-4, 1e-2, monkeypatch=monkeypatch)


def test_dropout_mask_statistics(dev):
    """The hashed mask: keep rate 1 - p, kept value 1 / (1 - p), different seeds decorrelated, and
    bit-identical to the oracle's restatement."""
    from speech2text_amd import conf_kernels as ck
    n, p = 1 << 20, 0.1
    ones = torch.ones(n, device=dev)
    m1 = ck.dropout_add(None, ones, 1.0, p, 12345).cpu()
    m2 = ck.dropout_add(None, ones, 1.0, p, 12346).cpu()
    assert torch.equal(m1, OC.keep_scale(12345, (n,), p))
    keep = (m1 > 0).float()
    assert abs(keep.mean().item() - (1 - p)) < 2e-3
    assert torch.all((m1 == 0) | ((m1 - 1 / (1 - p)).abs() < 1e-6))
    both = ((m1 > 0) & (m2 > 0)).float().mean().item()
    assert abs(both - (1 - p) ** 2) < 3e-3
    # neighbouring elements are independent
    assert abs((keep[1:] * keep[:-1]).mean().item() - (1 - p) ** 2) < 3e-3
    x = torch.randn(n, device=dev)
    y = torch.randn(n, device=dev)
    out = ck.dropout_add(x, y, 0.5, p, 12345).cpu()
    assert torch.allclose(out, x.cpu() + 0.5 * y.cpu() * m1, atol=1e-6)
    h = torch.randn(n, device=dev)
    a = ck.silu_fwd(h, p, 777).cpu()
    assert torch.allclose(a, torch.nn.functional.silu(h.cpu()) * OC.keep_scale(777, (n,), p), atol=1e-6)


def test_layer_executor_vs_oracle_c2_dims(dev):
    """C2 dims: 12 layers, d=256, 4 heads (dh 64), ffn 2048, k=31; 2 x 5 s utterances."""
    cfg = dict(input_dim=256, num_heads=4, ffn_dim=2048, num_layers=12,
               depthwise_conv_kernel_size=31, dropout=0.0, output_dim=256)
    _stack_vs_oracle(dev, cfg, 2, 498, torch.tensor([498, 401]), 2e-3, 2e-2, seed=1)


def test_shapes_outside_the_kernels_rules_use_torch_device_ops(dev):
    """Row lengths that are not multiples of 4 / head widths other than 16, 32, 64 / a BatchNorm
    with momentum=None / an empty LSTM sequence: the wrappers fall back to torch's DEVICE kernels
    (or handle the case) instead of failing with -2."""
    from speech2text_amd import conf_kernels as ck
    torch.manual_seed(0)
    ln = torch.nn.LayerNorm(6).to(dev)
    x = torch.randn(5, 7, 6, device=dev, requires_grad=True)
    torch.testing.assert_close(ck.layer_norm(x, ln), F.layer_norm(x, (6,), ln.weight, ln.bias, ln.eps))
    y = torch.randn(3, 7, device=dev)
    torch.testing.assert_close(ck.silu(y), F.silu(y))
    T, B, H, dh = 9, 2, 4, 36
    qkv = torch.randn(T, B, 3 * H * dh, device=dev)
    lens = torch.tensor([9, 5], device=dev)
    o = ck.mhsa(qkv, lens, H)
    q, k, v = (t.reshape(T, B, H, dh).permute(1, 2, 0, 3).double() for t in qkv.chunk(3, dim=-1))
    sc = q @ k.transpose(-1, -2) / math.sqrt(dh)
    sc = sc.masked_fill(torch.arange(T, device=dev)[None, None, None, :] >= lens[:, None, None, None], -1e30)
    ref = (sc.softmax(-1) @ v).permute(2, 0, 1, 3).reshape(T, B, H * dh)
    torch.testing.assert_close(o.double(), ref, atol=1e-5, rtol=1e-4)
    # cumulative-average BatchNorm (momentum=None), two batches
    bn = torch.nn.BatchNorm1d(8, momentum=None).to(dev)
    ref_bn = torch.nn.BatchNorm1d(8, momentum=None).to(dev)
    for _ in range(2):
        xb = torch.randn(12, 8, device=dev) * 2 + 1
        ck.batchnorm_silu(xb, bn)
        ref_bn(xb)
    torch.testing.assert_close(bn.running_mean, ref_bn.running_mean, atol=1e-6, rtol=1e-5)
    torch.testing.assert_close(bn.running_var, ref_bn.running_var, atol=1e-6, rtol=1e-5)
    # empty sequence: final state = initial state
    Hh = 8
    h0, c0 = torch.randn(2, Hh, device=dev), torch.randn(2, Hh, device=dev)
    hs, hT, cT = ck.lnlstm(torch.zeros(0, 2, 4 * Hh, device=dev), torch.randn(4 * Hh, Hh, device=dev),
                           torch.nn.Identity(), torch.nn.Identity(), h0, c0)
    assert hs.shape[0] == 0
    torch.testing.assert_close(hT, h0)
    torch.testing.assert_close(cT, c0)
