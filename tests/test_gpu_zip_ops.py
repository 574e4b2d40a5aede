"""GPU parity of individual zipformer HIP ops against the oracle's plain-torch statements."""
import numpy as np
import pytest
import torch

from oracle import zipformer as Z

pytestmark = pytest.mark.gpu


class _Conv(torch.nn.Module):
    def __init__(self, C, K):
        super().__init__()
        Kh = (K + 1) // 2
        self.kernel_size = K
        self.causal_conv = torch.nn.Conv1d(C, C, Kh, groups=C)
        self.chunkwise_conv = torch.nn.Conv1d(C, C, K, groups=C, padding=K // 2)
        self.chunkwise_conv_scale = torch.nn.Parameter(torch.randn(2, C, K) * 0.3)


@pytest.mark.parametrize("T,B,C,K,chunk", [(200, 3, 192, 31, -1), (130, 2, 96, 31, 32),
                                           (77, 4, 70, 15, 16), (50, 2, 64, 7, 8),
                                           (20, 2, 33, 31, -1), (64, 1, 64, 15, 4),
                                           (150, 7, 96, 31, -1), (90, 6, 64, 15, 16),
                                           (100, 2, 64, 15, 2), (96, 2, 64, 31, 12), (140, 2, 64, 31, 64)])
def test_glu_chunk_causal_dwconv(dev, monkeypatch, T, B, C, K, chunk):
    from speech2text_amd import zip_kernels as zk
    torch.manual_seed(0)
    if B >= 6:
        monkeypatch.setenv("S2T_CONV_BLOCKS", "4")     # several utterances per workgroup
    conv = _Conv(C, K)
    u = torch.randn(T, B, 2 * C)
    lens = torch.randint(T // 2, T + 1, (B,)); lens[0] = T
    mask = torch.arange(T).unsqueeze(0) >= lens.unsqueeze(1)
    wts = torch.randn(T, B, C)
    # oracle
    uc = u.clone().requires_grad_(True)
    sd = {"p.causal_conv.weight": conv.causal_conv.weight, "p.causal_conv.bias": conv.causal_conv.bias,
          "p.chunkwise_conv.weight": conv.chunkwise_conv.weight,
          "p.chunkwise_conv.bias": conv.chunkwise_conv.bias,
          "p.chunkwise_conv_scale": conv.chunkwise_conv_scale}
    x = uc[..., :C] * torch.sigmoid(uc[..., C:])
    x = x.permute(1, 2, 0).masked_fill(mask.unsqueeze(1), 0.0)
    yo = Z.chunk_causal_dwconv(sd, "p.", x, chunk, K).permute(2, 0, 1)
    (yo * wts).sum().backward()
    ref = {n: p.grad.clone() for n, p in conv.named_parameters()}
    for p in conv.parameters():
        p.grad = None
    # HIP
    conv_g = _Conv(C, K).to(dev)
    conv_g.load_state_dict(conv.state_dict())
    ug = u.to(dev).requires_grad_(True)
    y = zk.glu_chunk_causal_dwconv(ug, C, mask.to(dev), conv_g, chunk)
    (y * wts.to(dev)).sum().backward()
    np.testing.assert_allclose(y.detach().cpu().numpy(), yo.detach().numpy(), atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(ug.grad.cpu().numpy(), uc.grad.numpy(), atol=2e-5, rtol=1e-4)
    for n, p in conv_g.named_parameters():
        r = ref[n].numpy()
        np.testing.assert_allclose(p.grad.cpu().numpy(), r, atol=2e-4 * max(1.0, np.abs(r).max()),
                                   rtol=1e-3, err_msg=n)


def test_plain_depthwise_conv1d(dev):
    from speech2text_amd import zip_kernels as zk
    torch.manual_seed(1)
    T, B, C, K = 90, 3, 80, 15
    conv = torch.nn.Conv1d(C, C, K, groups=C, padding=K // 2)
    u = torch.randn(T, B, 2 * C)
    uc = u.clone().requires_grad_(True)
    x = (uc[..., :C] * torch.sigmoid(uc[..., C:])).permute(1, 2, 0)
    yo = conv(x).permute(2, 0, 1)
    yo.sum().backward()
    cg = torch.nn.Conv1d(C, C, K, groups=C, padding=K // 2).to(dev)
    cg.load_state_dict(conv.state_dict())
    ug = u.to(dev).requires_grad_(True)
    y = zk.glu_chunk_causal_dwconv(ug, C, None, cg, -1)
    y.sum().backward()
    np.testing.assert_allclose(y.detach().cpu().numpy(), yo.detach().numpy(), atol=2e-5)
    np.testing.assert_allclose(ug.grad.cpu().numpy(), uc.grad.numpy(), atol=2e-5)
    np.testing.assert_allclose(cg.weight.grad.cpu().numpy(), conv.weight.grad.numpy(), atol=5e-4, rtol=1e-3)
    np.testing.assert_allclose(cg.bias.grad.cpu().numpy(), conv.bias.grad.numpy(), atol=5e-4, rtol=1e-3)


@pytest.mark.parametrize("is_l", [True, False])
def test_swoosh_and_biasnorm(dev, is_l):
    from speech2text_amd import zip_kernels as zk
    torch.manual_seed(2)
    x = torch.randn(37, 5, 203) * 4
    xc = x.clone().requires_grad_(True)
    yo = (Z.swoosh_l if is_l else Z.swoosh_r)(xc)
    yo.sum().backward()
    xg = x.to(dev).requires_grad_(True)
    y = zk.swoosh(xg, is_l)
    y.sum().backward()
    np.testing.assert_allclose(y.detach().cpu().numpy(), yo.detach().numpy(), atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xc.grad.numpy(), atol=2e-6, rtol=1e-5)
    bias = torch.randn(203) * 0.1
    ls = torch.tensor(0.7)
    bc, lc = bias.clone().requires_grad_(True), ls.clone().requires_grad_(True)
    xc = x.clone().requires_grad_(True)
    w = torch.randn_like(x)
    yo = Z.bias_norm(xc, bc, lc, Z.Ctl(False))
    (yo * w).sum().backward()
    bg, lg = bias.to(dev).requires_grad_(True), ls.to(dev).requires_grad_(True)
    xg = x.to(dev).requires_grad_(True)
    y = zk.bias_norm(xg, bg, lg)
    (y * w.to(dev)).sum().backward()
    np.testing.assert_allclose(y.detach().cpu().numpy(), yo.detach().numpy(), atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xc.grad.numpy(), atol=1e-5, rtol=1e-4)
    np.testing.assert_allclose(bg.grad.cpu().numpy(), bc.grad.numpy(), atol=1e-3, rtol=1e-3)
    np.testing.assert_allclose(lg.grad.item(), lc.grad.item(), rtol=1e-4)


@pytest.mark.parametrize("B,T,D", [(3, 50, 192), (2, 33, 70), (5, 7, 256)])
def test_layout_carrying_passes_keep_values_and_save_the_copy(dev, B, T, D):
    """bias_norm_time_major (frontend output stored (T,B,D)) and simple_downsample(batch_major=True)
    (encoder output stored (B,T',D)): same values and gradients as the plain forms followed by a
    transpose, and the transposed result IS contiguous (no copy downstream)."""
    from speech2text_amd import zip_kernels as zk
    g = torch.Generator().manual_seed(B * T + D)
    x = (torch.randn(B, T, D, generator=g) * 2).to(dev)
    bias = (torch.randn(D, generator=g) * 0.1).to(dev)
    ls = torch.tensor(0.3, device=dev)
    w = torch.randn(T, B, D, generator=g).to(dev)
    outs = []
    for tm in (False, True):
        xg, bg, lg = (t.clone().requires_grad_(True) for t in (x, bias, ls))
        y = (zk.bias_norm_time_major if tm else zk.bias_norm)(xg, bg, lg)
        assert y.shape == (B, T, D)
        yt = y.transpose(0, 1)
        assert yt.is_contiguous() == tm
        (yt * w).sum().backward()
        outs.append((y.detach().clone(), xg.grad, bg.grad, lg.grad))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    torch.testing.assert_close(outs[0][2], outs[1][2], atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(outs[0][3], outs[1][3], atol=1e-3, rtol=1e-4)
    # the encoder's last downsample
    src = torch.randn(T, B, D, generator=g).to(dev)
    dsb = torch.randn(2, generator=g).to(dev)
    dT = (T + 1) // 2
    w2 = torch.randn(B, dT, D, generator=g).to(dev)
    outs = []
    for bm in (False, True):
        sg, bb = src.clone().requires_grad_(True), dsb.clone().requires_grad_(True)
        y = zk.simple_downsample(sg, bb, 2, batch_major=bm)
        assert y.shape == (dT, B, D)
        yb = y.transpose(0, 1)
        assert yb.is_contiguous() == bm
        (yb * w2).sum().backward()
        outs.append((y.detach().clone(), sg.grad, bb.grad))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    torch.testing.assert_close(outs[0][2], outs[1][2], atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("T,B,D,masked", [(50, 3, 192, True), (33, 2, 256, False), (20, 5, 68, True),
                                         (9, 1, 512, True)])
def test_norm_and_bypass_in_one_pass(dev, T, B, D, masked):
    """s2t_norm_bypass_fwd / _bwd (the end of a zipformer layer: BiasNorm, then the bypass, then the
    stack's feature mask) against the separate entry points they replace: outputs and the bypass's data
    gradient bit for bit, BiasNorm's data gradient and the per-channel parameter gradients to summation
    order."""
    import ctypes
    from speech2text_amd import _native as Nt
    L, st = Nt.lib(), Nt.stream()
    g = torch.Generator().manual_seed(T * D)
    R = T * B
    x = (torch.randn(R, D, generator=g) * 2).to(dev)
    orig = torch.randn(R, D, generator=g).to(dev)
    bias = (torch.randn(D, generator=g) * 0.1).to(dev)
    ls = torch.tensor([0.3], device=dev)
    k = torch.rand(D, generator=g).to(dev)
    fm = (torch.rand(B, D, generator=g) > 0.3).float().to(dev) if masked else None
    gy = torch.randn(R, D, generator=g).to(dev)
    e = lambda *shape: torch.empty(*shape, device=dev)                         # noqa: E731
    z = lambda *shape: torch.zeros(*shape, device=dev)                         # noqa: E731
    # the separate passes
    x10, sc0, out0 = e(R, D), e(R), e(R, D)
    assert L.s2t_biasnorm_fwd(Nt.fp(x), Nt.fp(bias), Nt.fp(ls), R, D, Nt.fp(x10), Nt.fp(sc0), st) == 0
    if masked:
        assert L.s2t_bypass_fwd_mask(Nt.fp(orig), Nt.fp(x10), Nt.fp(k), Nt.fp(fm), B, R, D, Nt.fp(out0), st) == 0
    else:
        assert L.s2t_bypass_fwd(Nt.fp(orig), Nt.fp(x10), Nt.fp(k), R, D, Nt.fp(out0), st) == 0
    d0, g10, dk0, dx0, db0, dl0 = e(R, D), e(R, D), z(D), e(R, D), z(D), z(1)
    if masked:
        assert L.s2t_bypass_bwd_mask(Nt.fp(orig), Nt.fp(x10), Nt.fp(k), Nt.fp(gy), Nt.fp(fm), B, R, D, Nt.fp(d0),
                                     Nt.fp(g10), Nt.fp(dk0), st) == 0
    else:
        assert L.s2t_bypass_bwd(Nt.fp(orig), Nt.fp(x10), Nt.fp(k), Nt.fp(gy), R, D, Nt.fp(d0), Nt.fp(g10),
                                Nt.fp(dk0), st) == 0
    assert L.s2t_biasnorm_bwd(Nt.fp(x), Nt.fp(bias), Nt.fp(sc0), Nt.fp(g10), R, D, Nt.fp(dx0), Nt.fp(db0),
                              Nt.fp(dl0), st) == 0
    # the fused pair
    sc1, out1 = e(R), e(R, D)
    assert L.s2t_norm_bypass_fwd(Nt.fp(x), Nt.fp(bias), Nt.fp(ls), Nt.fp(orig), Nt.fp(k), Nt.fp(fm), B, R, D,
                                 Nt.fp(out1), Nt.fp(sc1), st) == 0
    d1, dk1, dx1, db1, dl1 = e(R, D), z(D), e(R, D), z(D), z(1)
    assert L.s2t_norm_bypass_bwd(Nt.fp(x), Nt.fp(bias), Nt.fp(sc1), Nt.fp(orig), Nt.fp(k), Nt.fp(gy), Nt.fp(fm),
                                 B, R, D, Nt.fp(dx1), Nt.fp(d1), Nt.fp(dk1), Nt.fp(db1), Nt.fp(dl1), st) == 0
    assert torch.equal(sc0, sc1) and torch.equal(out0, out1)
    assert torch.equal(d0, d1)
    # (the 16-byte form of the fused backward takes the two sums over a row in another order: last bits)
    assert (dx0 - dx1).abs().max().item() <= 2e-6 * max(1.0, dx0.abs().max().item())
    for a, b in ((dk0, dk1), (db0, db1), (dl0, dl1)):
        assert (a - b).abs().max().item() <= 2e-5 * max(1.0, a.abs().max().item())


@pytest.mark.parametrize("T,B,C,K,chunk", [(200, 3, 192, 31, -1), (77, 4, 70, 15, 16), (64, 1, 64, 15, 4)])
def test_conv_module_activation_as_second_output(dev, T, B, C, K, chunk):
    """s2t_zipconv_fwd_act: y as s2t_zipconv_fwd writes it, and SwooshR / SwooshL of y equal to the
    separate activation pass bit for bit (16-byte and scalar store paths, chunked and not)."""
    from speech2text_amd import zip_kernels as zk
    torch.manual_seed(T + C)
    conv = _Conv(C, K).to(dev)
    u = torch.randn(T, B, 2 * C, device=dev)
    lens = torch.randint(T // 2, T + 1, (B,)); lens[0] = T
    m8 = (torch.arange(T).unsqueeze(0) >= lens.unsqueeze(1)).to(torch.uint8).to(dev)
    cp = zk.conv_params(conv, T, chunk)
    with torch.no_grad():
        y0 = zk.zipconv_forward(u, C, m8, *cp)
        for swl in (False, True):
            y, a = zk.zipconv_forward(u, C, m8, *cp, act=swl)
            assert torch.equal(y, y0)
            assert torch.equal(a, zk.swoosh_forward(y0.view(T * B, C), swl).view(T, B, C))


@pytest.mark.parametrize("rows,D", [(37 * 5, 192), (1031, 256), (4, 8), (3, 64), (257, 200), (130, 384),
                                    (66, 512), (9000, 192)])
def test_biasnorm_backward_ragged_shapes(dev, rows, D):
    """s2t_biasnorm_bwd against the oracle: row counts that are not multiples of four, partial last
    column group, wide rows"""
    from speech2text_amd import zip_kernels as zk
    torch.manual_seed(rows + D)
    x = torch.randn(rows, D) * 2
    bias = torch.randn(D) * 0.1
    ls = torch.tensor(0.3)
    w = torch.randn_like(x)
    xc, bc, lc = x.clone().requires_grad_(True), bias.clone().requires_grad_(True), ls.clone().requires_grad_(True)
    (Z.bias_norm(xc, bc, lc, Z.Ctl(False)) * w).sum().backward()
    xg, bg, lg = (t.to(dev).requires_grad_(True) for t in (x, bias, ls))
    (zk.bias_norm(xg, bg, lg) * w.to(dev)).sum().backward()
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xc.grad.numpy(), atol=1e-5, rtol=1e-4)
    tol = 1e-3 * max(1.0, float(bc.grad.abs().max()))
    np.testing.assert_allclose(bg.grad.cpu().numpy(), bc.grad.numpy(), atol=tol, rtol=1e-3)
    np.testing.assert_allclose(lg.grad.item(), lc.grad.item(), rtol=2e-4, atol=1e-3)


def _attn_ref(qkp, pos, H, qd, pd, amask, kpm):
    T, B, _ = qkp.shape
    q = qkp[..., :H * qd].reshape(T, B, H, qd).permute(2, 1, 0, 3)
    k = qkp[..., H * qd:2 * H * qd].reshape(T, B, H, qd).permute(2, 1, 3, 0)
    p = qkp[..., 2 * H * qd:].reshape(T, B, H, pd).permute(2, 1, 0, 3)
    s = torch.matmul(q, k)
    if pos is not None:
        pe = pos.reshape(1, 2 * T - 1, H, pd).permute(2, 0, 3, 1)
        ps = torch.matmul(p, pe)
        idx = (T - 1) - torch.arange(T).unsqueeze(1) + torch.arange(T).unsqueeze(0)
        s = s + ps[:, :, torch.arange(T).unsqueeze(1), idx]
    if amask is not None:
        s = s.masked_fill(amask, -1000)
    if kpm is not None:
        s = s.masked_fill(kpm.unsqueeze(1), -1000)
    return s.softmax(dim=-1)


@pytest.mark.parametrize("T,B,H,qd,pd,dvs,use_dw0,use_am", [
    (50, 2, 4, 8, 4, (12,), False, True), (130, 3, 4, 32, 4, (12, 12), True, False),
    (530, 2, 2, 32, 4, (16, 16), True, False), (77, 2, 4, 24, 4, (12, 12), True, True),
    (1, 2, 2, 8, 4, (12,), True, False), (33, 1, 8, 4, 4, (4, 12), False, False),
    # the C3 stacks' shapes, ragged last passes, masks, pd 8
    (495, 2, 4, 32, 4, (12, 12), True, False), (248, 2, 4, 32, 4, (12, 12), True, True),
    (124, 2, 4, 32, 4, (12, 12), True, False), (300, 2, 8, 32, 8, (12, 12), False, True),
    (512, 1, 2, 16, 3, (16, 16), True, False), (97, 3, 2, 32, 4, (12,), True, True)])
def test_relpos_attention_backward_factored(dev, T, B, H, qd, pd, dvs, use_dw0, use_am):
    """s2t_relpos_attn_bwd with dW given as factors (dO_c, V_c) + head-0 term + delta, against the
    float64 autograd of the composed op with the same dW materialised.  Two kernels (query-owner +
    key-owner) at every length."""
    from speech2text_amd import zip_kernels as zk
    torch.manual_seed(T + 7)
    Dp = H * (2 * qd + pd)
    qkp = torch.randn(T, B, Dp, dtype=torch.float64) * 0.7
    pos = torch.randn(2 * T - 1, H * pd, dtype=torch.float64)
    lens = torch.randint(max(1, T // 2), T + 1, (B,)); lens[0] = T
    kpm = torch.arange(T).unsqueeze(0) >= lens.unsqueeze(1)
    amask = None
    if use_am:
        c = torch.arange(T) // 16
        amask = torch.logical_or(c.unsqueeze(0) > c.unsqueeze(1), c.unsqueeze(0) < c.unsqueeze(1) - 2)
    facs = [(torch.randn(T, B, H * dv, dtype=torch.float64), torch.randn(T, B, H * dv, dtype=torch.float64), dv)
            for dv in dvs]
    dW0 = torch.randn(B, T, T, dtype=torch.float64) if use_dw0 else None
    dWm = torch.zeros(H, B, T, T, dtype=torch.float64)
    for dO, V, dv in facs:
        dWm += torch.einsum("ibhd,jbhd->hbij", dO.view(T, B, H, dv), V.view(T, B, H, dv))
    if use_dw0:
        dWm[0] += dW0
    qc = qkp.clone().requires_grad_(True); pc = pos.clone().requires_grad_(True)
    Wr = _attn_ref(qc, pc, H, qd, pd, amask, kpm)
    (Wr * dWm).sum().backward()
    f = lambda t: None if t is None else t.float().to(dev).contiguous()
    qg, pg = f(qkp), f(pos)
    k8 = kpm.to(torch.uint8).to(dev); a8 = None if amask is None else amask.to(torch.uint8).to(dev)
    W = zk.relpos_attention_weights(qg, pg, H, qd, pd, None if amask is None else amask.to(dev), kpm.to(dev))
    delta = (W.double() * dWm.to(dev)).sum(-1).float().contiguous()
    pairs = [(f(dO), f(V), None, dv) for dO, V, dv in facs]
    dq, dp = zk._attn_bwd_call(qg, pg, k8, a8, H, qd, pd, W, None, f(dW0), pairs, delta)
    gq, gp = qc.grad.numpy(), pc.grad.numpy()
    np.testing.assert_allclose(dq.cpu().numpy(), gq, atol=3e-5 * max(1.0, np.abs(gq).max()), rtol=2e-3)
    np.testing.assert_allclose(dp.cpu().numpy(), gp, atol=3e-5 * max(1.0, np.abs(gp).max()), rtol=2e-3)
    # mixed: one factor pair + the rest materialised, delta recomputed by the library
    rest = dWm.clone()
    dO, V, dv = facs[0]
    rest -= torch.einsum("ibhd,jbhd->hbij", dO.view(T, B, H, dv), V.view(T, B, H, dv))
    dq2, dp2 = zk._attn_bwd_call(qg, pg, k8, a8, H, qd, pd, W, f(rest), None, pairs[:1], delta)
    np.testing.assert_allclose(dq2.cpu().numpy(), gq, atol=3e-5 * max(1.0, np.abs(gq).max()), rtol=2e-3)
    np.testing.assert_allclose(dp2.cpu().numpy(), gp, atol=3e-5 * max(1.0, np.abs(gp).max()), rtol=2e-3)


@pytest.mark.parametrize("T,B,H,qd,pd,use_pos,use_am", [
    (50, 2, 4, 8, 4, True, True), (130, 3, 8, 32, 4, True, False), (64, 2, 2, 16, 4, True, True),
    (530, 2, 2, 32, 4, True, False), (1, 2, 2, 8, 4, True, False), (77, 2, 4, 24, 4, False, True),
    (200, 2, 4, 32, 8, True, True),
    # the MFMA forward: C3 stack shapes (T, H, qd of stacks 0..3), both strip layouts, edges
    (495, 3, 4, 32, 4, True, False), (495, 2, 4, 32, 4, True, True), (248, 2, 4, 32, 4, True, False),
    (124, 2, 4, 32, 4, True, True), (62, 3, 8, 32, 4, True, False), (257, 2, 2, 16, 3, True, True),
    (512, 1, 2, 32, 4, False, False), (33, 2, 2, 8, 2, True, False), (256, 2, 2, 32, 4, True, False)])
def test_relpos_attention_weights(dev, T, B, H, qd, pd, use_pos, use_am):
    from speech2text_amd import zip_kernels as zk
    torch.manual_seed(T)
    Dp = H * (2 * qd + pd)
    qkp = torch.randn(T, B, Dp, dtype=torch.float64) * 0.7
    pos = torch.randn(2 * T - 1, H * pd, dtype=torch.float64) if use_pos else None
    lens = torch.randint(max(1, T // 2), T + 1, (B,)); lens[0] = T
    kpm = torch.arange(T).unsqueeze(0) >= lens.unsqueeze(1)
    amask = None
    if use_am:
        c = torch.arange(T) // 16
        amask = torch.logical_or(c.unsqueeze(0) > c.unsqueeze(1), c.unsqueeze(0) < c.unsqueeze(1) - 2)
    wts = torch.randn(H, B, T, T, dtype=torch.float64)
    qc = qkp.clone().requires_grad_(True)
    pc = pos.clone().requires_grad_(True) if use_pos else None
    Wr = _attn_ref(qc, pc, H, qd, pd, amask, kpm)
    (Wr * wts).sum().backward()
    qg = qkp.float().to(dev).requires_grad_(True)
    pg = pos.float().to(dev).requires_grad_(True) if use_pos else None
    W = zk.relpos_attention_weights(qg, pg, H, qd, pd, None if amask is None else amask.to(dev),
                                    kpm.to(dev))
    (W * wts.float().to(dev)).sum().backward()
    np.testing.assert_allclose(W.detach().cpu().numpy(), Wr.detach().numpy(), atol=2e-6, rtol=2e-4)
    gq = qc.grad.numpy()
    np.testing.assert_allclose(qg.grad.cpu().numpy(), gq, atol=3e-5 * max(1.0, np.abs(gq).max()), rtol=2e-3)
    if use_pos:
        gp = pc.grad.numpy()
        np.testing.assert_allclose(pg.grad.cpu().numpy(), gp, atol=3e-5 * max(1.0, np.abs(gp).max()), rtol=2e-3)


@pytest.mark.parametrize("R,Nf,Mf,bias,stride_pad", [
    (4096, 192, 192, True, 0), (31680, 576, 192, True, 0), (3001, 64, 130, False, 0),
    (2049, 48, 256, True, 0), (5000, 500, 512, True, 0), (1500, 2, 2, True, 0),
    (4096, 128, 64, True, 64), (15872, 256, 768, True, 0)])
def test_linear_wgrad(dev, R, Nf, Mf, bias, stride_pad, request):
    """s2t_linear_wgrad against the float64 product (weight / bias gradient of nn.Linear).  A
    kernel-level bound of the SIX-product arithmetic (2e-6 sqrt(R) x 4 absolute on unit-variance
    operands): the arithmetic is pinned to bf16x3 here; the two-piece mode's own bound is asserted
    by tests/test_gpu_gemm.py under both arithmetics."""
    from speech2text_amd import zip_kernels as zk, _native as N
    assert N.lib().s2t_gemm_arith_set(3) == 0
    request.addfinalizer(lambda: N.lib().s2t_gemm_arith_set(0))
    torch.manual_seed(R + Nf)
    gfull = torch.randn(R, Nf + stride_pad, device=dev)
    afull = torch.randn(R, Mf + stride_pad, device=dev)
    g, a = gfull[:, :Nf], afull[:, :Mf]
    dW = torch.full((Nf, Mf), float("nan"), device=dev)
    db = torch.full((Nf,), float("nan"), device=dev) if bias else None
    ws = torch.empty(N.lib().s2t_linear_wgrad_workspace_floats(R, Nf, Mf), device=dev)
    gp, ap = N.raw(g, torch.float32), N.raw(a, torch.float32)
    N.check(N.lib().s2t_linear_wgrad(gp, g.stride(0), ap, a.stride(0), R, Nf, Mf, N.fp(dW),
                                     N.fp(db), 0, N.fp(ws), N.stream()), "s2t_linear_wgrad")
    ref = g.double().t() @ a.double()
    tol = 2e-6 * R ** 0.5 * 4
    np.testing.assert_allclose(dW.cpu().numpy(), ref.cpu().numpy(), atol=tol, rtol=1e-5)
    if bias:
        np.testing.assert_allclose(db.cpu().numpy(), g.double().sum(0).cpu().numpy(), atol=tol, rtol=1e-5)
    # accumulate=1 adds on top
    N.check(N.lib().s2t_linear_wgrad(gp, g.stride(0), ap, a.stride(0), R, Nf, Mf, N.fp(dW),
                                     N.fp(db), 1, N.fp(ws), N.stream()), "s2t_linear_wgrad")
    np.testing.assert_allclose(dW.cpu().numpy(), 2 * ref.cpu().numpy(), atol=2 * tol, rtol=1e-5)
    # the autograd wrapper
    x = torch.randn(7, R // 7, Mf, device=dev, requires_grad=True)
    w = torch.randn(Nf, Mf, device=dev, requires_grad=True)
    b = torch.randn(Nf, device=dev, requires_grad=True) if bias else None
    gy = torch.randn(7, R // 7, Nf, device=dev)
    zk.linear(x, w, b).backward(gy)
    x2, w2 = x.detach().clone().requires_grad_(True), w.detach().clone().requires_grad_(True)
    b2 = b.detach().clone().requires_grad_(True) if bias else None
    torch.nn.functional.linear(x2, w2, b2).backward(gy)
    np.testing.assert_allclose(w.grad.cpu().numpy(), w2.grad.cpu().numpy(), atol=2 * tol, rtol=1e-4)
    np.testing.assert_allclose(x.grad.cpu().numpy(), x2.grad.cpu().numpy(), atol=1e-4, rtol=1e-4)
    if bias:
        np.testing.assert_allclose(b.grad.cpu().numpy(), b2.grad.cpu().numpy(), atol=2 * tol, rtol=1e-4)


@pytest.mark.parametrize("N_,H,W,C,K", [(3, 37, 19, 128, 7), (2, 9, 5, 70, 7), (5, 130, 19, 64, 3),
                                        (2, 12, 8, 16, 3),    # (even width: the gather-form dgrad's last column)
                                        (1, 3, 5, 8, 3)])     # (one patch row per stride-2 conv: fallback paths)
def test_dwconv2d_and_conv3x3_nhwc_vs_torch(dev, N_, H, W, C, K):
    """Channel-last frontend convolutions (zip_front.hip: depthwise stencil, its two-stage weight
    gradient, the col2im gather of the 3x3 conv) against torch's NCHW convolutions in fp64."""
    from speech2text_amd import zip_kernels as zk
    g = torch.Generator().manual_seed(2)
    x = torch.randn(N_, H, W, C, generator=g)
    w = torch.randn(C, 1, K, K, generator=g) * 0.2
    b = torch.randn(C, generator=g)
    wts = torch.randn(N_, H, W, C, generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = torch.nn.functional.conv2d(xr.permute(0, 3, 1, 2), wr, br, padding=K // 2, groups=C)
    (yr.permute(0, 2, 3, 1) * wts.double()).sum().backward()
    xg, wg, bg = (t.to(dev).requires_grad_(True) for t in (x, w, b))
    y = zk.dwconv2d_nhwc(xg, wg, bg)
    (y * wts.to(dev)).sum().backward()
    torch.testing.assert_close(y.detach().cpu().double(), yr.permute(0, 2, 3, 1).detach(), atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(xg.grad.cpu().double(), xr.grad, atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(wg.grad.cpu().double(), wr.grad, atol=2e-3, rtol=2e-4)
    torch.testing.assert_close(bg.grad.cpu().double(), br.grad, atol=2e-3, rtol=2e-4)
    # first subsampling conv: 1 -> 8 channels, padding (0, 1), direct stencil kernels
    x1 = torch.randn(N_, H, W, 1, generator=g)
    w1 = torch.randn(8, 1, 3, 3, generator=g) * 0.3
    b1 = torch.randn(8, generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x1, w1, b1))
    yr = torch.nn.functional.conv2d(xr.permute(0, 3, 1, 2), wr, br, padding=(0, 1))
    wt1 = torch.randn(yr.shape, generator=g, dtype=torch.float64)
    (yr * wt1).sum().backward()
    xg, wg, bg = (t.to(dev).requires_grad_(True) for t in (x1, w1, b1))
    y = zk.conv3x3_nhwc(xg, wg, bg, (1, 1), pad_w=1)
    (y * wt1.permute(0, 2, 3, 1).float().to(dev)).sum().backward()
    torch.testing.assert_close(y.detach().cpu().double(), yr.permute(0, 2, 3, 1).detach(), atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(xg.grad.cpu().double(), xr.grad, atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(wg.grad.cpu().double(), wr.grad, atol=2e-3, rtol=2e-4)
    torch.testing.assert_close(bg.grad.cpu().double(), br.grad, atol=2e-3, rtol=2e-4)
    # second subsampling conv: 8 -> 32 channels, stride 2, direct forward / input-gradient kernels
    if H >= 9 and W >= 5:
        x2 = torch.randn(N_, H, W, 8, generator=g)
        w2 = torch.randn(32, 8, 3, 3, generator=g) * 0.2
        b2 = torch.randn(32, generator=g)
        xr, wr, br = (t.double().requires_grad_(True) for t in (x2, w2, b2))
        yr = torch.nn.functional.conv2d(xr.permute(0, 3, 1, 2), wr, br, stride=2)
        wt2 = torch.randn(yr.shape, generator=g, dtype=torch.float64)
        (yr * wt2).sum().backward()
        xg, wg, bg = (t.to(dev).requires_grad_(True) for t in (x2, w2, b2))
        y = zk.conv3x3_nhwc(xg, wg, bg, (2, 2))
        (y * wt2.permute(0, 2, 3, 1).float().to(dev)).sum().backward()
        torch.testing.assert_close(y.detach().cpu().double(), yr.permute(0, 2, 3, 1).detach(), atol=1e-4, rtol=1e-4)
        torch.testing.assert_close(xg.grad.cpu().double(), xr.grad.permute(0, 1, 2, 3), atol=1e-4, rtol=1e-4)
        torch.testing.assert_close(wg.grad.cpu().double(), wr.grad, atol=2e-3, rtol=2e-4)
        torch.testing.assert_close(bg.grad.cpu().double(), br.grad, atol=2e-3, rtol=2e-4)
    # 3x3 conv: implicit-im2col MFMA GEMM forward / weight gradient (s2t_conv3x3_gemm), col2im
    # gather for the input gradient -- stride (1, 2) with >= 16 input channels: the gather-form GEMM by
    # column parity (s2t_gemm_x3p_map, no patch-space matrix); (6, 10): the materialised fallback
    for Cin, Cout, stride in ((8, 16, (1, 1)), (8, 16, (2, 2)), (8, 16, (1, 2)), (32, 128, (1, 2)),
                              (16, 48, (1, 2)), (6, 10, (1, 1))):
        x3 = torch.randn(N_, H, W, Cin, generator=g)
        w3 = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.2
        b3 = torch.randn(Cout, generator=g)
        xr, wr, br = (t.double().requires_grad_(True) for t in (x3, w3, b3))
        yr = torch.nn.functional.conv2d(xr.permute(0, 3, 1, 2), wr, br, stride=stride)
        wt3 = torch.randn(yr.shape, generator=g, dtype=torch.float64)
        (yr * wt3).sum().backward()
        xg, wg, bg = (t.to(dev).requires_grad_(True) for t in (x3, w3, b3))
        y = zk.conv3x3_nhwc(xg, wg, bg, stride)
        (y * wt3.permute(0, 2, 3, 1).float().to(dev)).sum().backward()
        torch.testing.assert_close(y.detach().cpu().double(), yr.permute(0, 2, 3, 1).detach(), atol=1e-4, rtol=1e-4)
        torch.testing.assert_close(xg.grad.cpu().double(), xr.grad, atol=1e-4, rtol=1e-4)
        torch.testing.assert_close(wg.grad.cpu().double(), wr.grad, atol=2e-3, rtol=2e-4)
        torch.testing.assert_close(bg.grad.cpu().double(), br.grad, atol=2e-3, rtol=2e-4)


@pytest.mark.parametrize("T,B,C,ds", [(37, 3, 64, 2), (64, 2, 96, 4), (101, 4, 128, 8), (5, 1, 32, 4),
                                      (9, 3, 33, 2), (495, 2, 256, 2)])
def test_downsample_and_upsampled_bypass_vs_torch(dev, T, B, C, ds):
    """zip_glue.hip SimpleDownsample and the fused SimpleUpsample + out_combiner bypass against the
    torch compositions of the reference's formulas (zipformer.py:1653-1719, 1523-1555), fp64."""
    from speech2text_amd import zip_kernels as zk
    g = torch.Generator().manual_seed(T)
    src = torch.randn(T, B, C, generator=g)
    bias = torch.randn(ds, generator=g)
    dT = (T + ds - 1) // ds
    wts = torch.randn(dT, B, C, generator=g)
    sr, br = src.double().requires_grad_(True), bias.double().requires_grad_(True)
    pad = dT * ds - T
    sp = torch.cat((sr, sr[T - 1:].expand(pad, B, C)), dim=0) if pad else sr
    yr = (sp.reshape(dT, ds, B, C) * br.softmax(0).reshape(1, ds, 1, 1)).sum(dim=1)
    (yr * wts.double()).sum().backward()
    sg, bg = src.to(dev).requires_grad_(True), bias.to(dev).requires_grad_(True)
    y = zk.simple_downsample(sg, bg, ds)
    (y * wts.to(dev)).sum().backward()
    torch.testing.assert_close(y.detach().cpu().double(), yr.detach(), atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(sg.grad.cpu().double(), sr.grad, atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(bg.grad.cpu().double(), br.grad, atol=1e-3, rtol=1e-4)
    # upsample + bypass: orig (T,B,C), low-rate src (dT,B,C)
    orig = torch.randn(T, B, C, generator=g)
    low = torch.randn(dT, B, C, generator=g)
    scale = torch.rand(C, generator=g)
    w2 = torch.randn(T, B, C, generator=g)
    o_r, l_r, s_r = (t.double().requires_grad_(True) for t in (orig, low, scale))
    upr = l_r.unsqueeze(1).expand(dT, ds, B, C).reshape(dT * ds, B, C)[:T]
    zr = o_r + (upr - o_r) * s_r
    (zr * w2.double()).sum().backward()
    o_g, l_g, s_g = (t.to(dev).requires_grad_(True) for t in (orig, low, scale))
    z = zk.bypass_upsampled(o_g, l_g, s_g, ds)
    (z * w2.to(dev)).sum().backward()
    torch.testing.assert_close(z.detach().cpu().double(), zr.detach(), atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(o_g.grad.cpu().double(), o_r.grad, atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(l_g.grad.cpu().double(), l_r.grad, atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(s_g.grad.cpu().double(), s_r.grad, atol=1e-3, rtol=1e-4)


@pytest.mark.parametrize("R,C,G", [(700, 192, 1), (1000, 128, 4), (333, 96, 2), (2000, 512, 1),
                                   (500, 256, 8), (260, 64, 1)])
@pytest.mark.parametrize("form", ["fwdpg", "fused", "sq", "pass"])
def test_whiten_backward_vs_oracle(dev, monkeypatch, R, C, G, form):
    """Whiten (scaling.py:949-1095): x^T x comes from the symmetric TN GEMM (only the 64x64 tiles on /
    above the diagonal with same-group pairs; cg = 48 straddles tiles), the metric kernel mirrors
    them; the backward term against the oracle's autograd-in-backward statement.  form: "fwdpg" (built
    late in round 6, measured slower in the step, off by default) = the penalty product x dcov taken in FORWARD too, with ||pg||^2 from its epilogue; backward =
    s2t_sumsq64 (||g||^2) + s2t_whiten_combine64; "fused" = dcov
    and its bf16 pieces taken in forward, backward = the penalty product on the pre-split-weight kernel
    with the two norms in its epilogue + the combining pass (s2t_whiten_prep / s2t_gemm_x3p_sq /
    s2t_whiten_combine64: the default);
    "sq" = the norms of (g, x dcov) from the NN product's epilogue (s2t_gemm_f32_sq); "pass" = from a
    pass over both tensors."""
    import random
    from speech2text_amd import zip_kernels as zkm
    from speech2text_amd.model.layer.scaling import Whiten
    monkeypatch.setattr(zkm, "_WHITEN_X3P", 2 if form in ("fused", "fwdpg") else 0)
    monkeypatch.setattr(zkm, "_WHITEN_FWD_PG", form == "fwdpg")
    monkeypatch.setattr(zkm, "_WHITEN_SQ", form == "sq")
    fused0 = int(N_lib().s2t_gemm_x3p_calls())
    torch.manual_seed(R + C)
    x = torch.randn(R, C) @ (torch.eye(C) + 0.3 * torch.randn(C, C))      # correlated channels
    x = x + 0.5 * torch.randn(C)
    w = torch.randn(R, C)
    monkeypatch.setattr(random, "random", lambda: 0.0)                     # the module fires
    ctl = Z.Ctl(training=True, rand=lambda: 0.0)
    xc = x.clone().requires_grad_(True)
    (Z.whiten(xc, ctl, G, 1.02, 0.05) * w).sum().backward()
    m = Whiten(num_groups=G, whitening_limit=1.02, prob=(0.025, 0.25), grad_scale=0.05).to(dev).train()
    for _ in range(2):                       # twice: the shared accumulator must come back clean
        xg = x.to(dev).requires_grad_(True)
        y = m(xg)
        (y * w.to(dev)).sum().backward()
        assert torch.equal(y.detach().cpu(), x)
        d = (xg.grad.cpu() - w).numpy()                                    # the shaping term only
        r = (xc.grad - w).numpy()
        assert np.abs(r).max() > 0, "metric below the limit: the test would be vacuous"
        np.testing.assert_allclose(d, r, atol=2e-3 * np.abs(r).max(), rtol=2e-3)
    # the fused form really ran (C % 8 == 0 everywhere here): one s2t_gemm_x3p launch per backward
    assert int(N_lib().s2t_gemm_x3p_calls()) - fused0 == (2 if form in ("fused", "fwdpg") else 0)


def test_whiten_backward_rank_one_input_stays_finite(dev, monkeypatch):
    """All rows parallel (metric = C, e.g. a freshly initialised projection): d metric / dx
    vanishes analytically and what the GEMM returns is rounding noise, which the reference (and
    the product) normalise to grad_scale |g|: the result must stay of that size, never huge."""
    import random
    from speech2text_amd.model.layer.scaling import Whiten
    torch.manual_seed(3)
    R, C = 990, 192
    x = torch.randn(R, 1) * torch.randn(1, C) + 0.25
    w = torch.randn(R, C)
    monkeypatch.setattr(random, "random", lambda: 0.0)
    m = Whiten(num_groups=1, whitening_limit=5.0, prob=(0.025, 0.25), grad_scale=0.01).to(dev).train()
    xg = x.to(dev).requires_grad_(True)
    (m(xg) * w.to(dev)).sum().backward()
    d = xg.grad.cpu() - w
    assert torch.isfinite(d).all()
    assert float(d.norm()) <= 0.1 * float(w.norm())         # reference: grad_scale |g| = 0.01 |g| of noise


@pytest.mark.parametrize("T,B,H,dv", [(495, 3, 4, 12), (496, 3, 4, 12), (248, 2, 4, 12), (5, 1, 2, 4), (3, 2, 1, 4), (124, 2, 8, 12), (62, 3, 4, 12),
                                      (200, 2, 2, 16), (130, 2, 3, 4), (77, 2, 4, 12), (64, 1, 1, 8)])
def test_attn_apply_both_ways(dev, T, B, H, dv):
    """s2t_attn_apply: out = W v per head (zipformer.py:2269) and its transpose (the value
    gradient), float4 path (dv % 4 == 0; rows unaligned when T % 4 != 0) and the general one,
    against fp64 matmuls."""
    from speech2text_amd import _native as Nt
    torch.manual_seed(T + dv)
    W = torch.rand(H, B, T, T).softmax(-1).to(dev).contiguous()
    v = torch.randn(T, B, H * dv).to(dev)
    out = torch.full_like(v, float("nan"))
    for tr in (0, 1):
        Nt.check(Nt.lib().s2t_attn_apply(Nt.fp(W), Nt.fp(v), T, B, H, dv, tr, Nt.fp(out), Nt.stream()),
                 "s2t_attn_apply")
        Wd = W.double().transpose(-1, -2) if tr else W.double()
        ref = torch.matmul(Wd, v.double().reshape(T, B, H, dv).permute(2, 1, 0, 3))
        ref = ref.permute(2, 1, 0, 3).reshape(T, B, H * dv)
        np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), atol=2e-5, rtol=1e-4)


def N_lib():
    from speech2text_amd import _native
    return _native.lib()


def _balancer_ref64(x, g, min_mean, max_mean, min_rms, max_rms, grad_scale, swoosh):
    """reference model/layer/scaling.py:741-789 in closed form (as zk.balancer_backward's torch path), fp64."""
    x, g = x.double(), g.double()
    if swoosh is not None:
        g = g * (torch.sigmoid(x - (4.0 if swoosh else 1.0)) - 0.08)
    n = x.shape[0]
    mean = x.mean(0, keepdim=True)
    uvar = (x * x).mean(0, keepdim=True)
    var = (uvar - mean * mean).clamp(min=1.0e-20)
    std, rms = var.sqrt(), uvar.clamp(min=1.0e-20).sqrt()
    m = mean / std
    s_m = torch.sign(m - m.clamp(min=min_mean, max=max_mean))
    s_r = -torch.sign((rms.clamp(min=min_rms, max=max_rms) / rms).log())
    a = s_m / n * (1.0 / std + mean * mean / (std * var))
    b = -s_m / n * mean / (std * var) + s_r / n / (rms * rms)
    lg_rms = (a * a + 2 * a * b * mean + b * b * uvar).clamp(min=0).sqrt().clamp(min=1.0e-20)
    return g + g.abs() * (a + b * x) * (grad_scale / lg_rms)


@pytest.mark.parametrize("rows,C", [(31680, 192), (15872, 576), (7936, 960), (3968, 1024), (1000, 100), (37, 256),
                                    (5000, 260), (300001, 8), (50001, 32), (4097, 16), (333, 4)])
@pytest.mark.parametrize("swoosh", [None, True])
def test_balancer_backward_vs_fp64_closed_form(dev, rows, C, swoosh):
    """s2t_balancer_bwd (reference model/layer/scaling.py:741-789 in closed form; two passes: column
    statistics, fused update) against the fp64 closed form: channels on both sides of every clamp
    (|mean| / std <= 8: the fp32 statistics' var = E[x^2] - mean^2 keeps four digits), row counts that
    leave ragged tails, channel counts that are no multiple of 64, few channels (C = 4 ... 32 contiguous:
    the flat 16-byte form the frontend's first convolutions take), operands at addresses that are no
    multiple of 16 bytes, a row-strided slice of a wider tensor, with and without the Swoosh derivative
    in front."""
    from speech2text_amd import zip_kernels as zk
    g0 = torch.Generator().manual_seed(rows + C)
    x = (torch.randn(rows, C, generator=g0) * torch.logspace(-0.7, 1.0, C) + torch.linspace(-1.5, 1.5, C)).to(dev)
    g = torch.randn(rows, C, generator=g0).to(dev)
    cfg = (-0.05, 0.6, 0.3, 4.0, 0.04)
    out = zk.balancer_backward(x, g, *cfg, 1, swoosh_l=swoosh)
    ref = _balancer_ref64(x, g, *cfg, swoosh)
    scale = ref.abs().max().item()
    upd = (out.double() - ref).abs().max().item()
    size = (ref - (g.double() * ((torch.sigmoid(x.double() - 4.0) - 0.08) if swoosh else 1.0))).abs().max().item()
    assert size > 1e-3 * scale                                               # the update is really there
    assert upd <= 2e-3 * size + 2e-6 * scale, (upd, size, scale)
    # the same values at an address that is not a multiple of 16
    bx = torch.empty(rows * C + 1, device=dev)
    bg = torch.empty(rows * C + 1, device=dev)
    xm, gm = bx[1:].view(rows, C), bg[1:].view(rows, C)
    xm.copy_(x)
    gm.copy_(g)
    assert xm.data_ptr() % 16 != 0
    out4 = zk.balancer_backward(xm, gm, *cfg, 1, swoosh_l=swoosh)
    assert (out4.double() - ref).abs().max().item() <= 2e-3 * size + 2e-6 * scale
    assert (out - out4).abs().max().item() <= 2e-3 * size + 2e-6 * scale
    # a row-strided slice of a wider tensor (the gate third of the nonlinear attention's projection)
    wide = torch.randn(rows, 3 * C, generator=g0).to(dev)
    wide[:, C:2 * C] = x
    outs = zk.balancer_backward(wide[:, C:2 * C], g, *cfg, 1, swoosh_l=swoosh)
    assert (outs.double() - ref).abs().max().item() <= 2e-3 * size + 2e-6 * scale


@pytest.mark.parametrize("T,B,C,up", [(495, 64, 256, 2), (495, 64, 256, 4), (495, 64, 192, 8), (248, 64, 384, 2),
                                      (131, 7, 64, 4)])
def test_upsampled_bypass_backward_16_byte_form_full_size(dev, T, B, C, up):
    """s2t_bypass_up_bwd at the C3 sizes -- the 16-byte form's multi-item loop, its capped grid (a lane
    keeps its channels only because the grid's stride is a multiple of C / 4: C = 192 / 384 are the
    cases where that needs a rounded grid) and the ragged last source frame -- against the closed
    form in fp64 on the device (zipformer.py:1253-1283, 1523-1555)."""
    from speech2text_amd import _native as Nt
    L, st = Nt.lib(), Nt.stream()
    g = torch.Generator().manual_seed(T + C + up)
    Ts = (T + up - 1) // up
    orig = torch.randn(T, B, C, generator=g).to(dev)
    src = torch.randn(Ts, B, C, generator=g).to(dev)
    k = torch.rand(C, generator=g).to(dev)
    gy = torch.randn(T, B, C, generator=g).to(dev)
    d_orig, d_src, d_k = torch.empty_like(orig), torch.empty_like(src), torch.zeros_like(k)
    assert L.s2t_bypass_up_bwd(Nt.fp(orig), Nt.fp(src), Nt.fp(k), Nt.fp(gy), up, T, B, C, Nt.fp(d_orig),
                               Nt.fp(d_src), Nt.fp(d_k), st) == 0
    gd, od, kd = gy.double(), orig.double(), k.double()
    sup = src.double().repeat_interleave(up, dim=0)[:T]
    r_orig = gd * (1.0 - kd)
    gpad = torch.cat((gd, torch.zeros(Ts * up - T, B, C, dtype=torch.float64, device=dev)), dim=0)
    r_src = (gpad * kd).reshape(Ts, up, B, C).sum(dim=1)
    r_k = (gd * (sup - od)).sum(dim=(0, 1))
    assert (d_orig.double() - r_orig).abs().max().item() <= 1e-6 * r_orig.abs().max().item()
    assert (d_src.double() - r_src).abs().max().item() <= 2e-6 * r_src.abs().max().item()
    assert (d_k.double() - r_k).abs().max().item() <= 2e-5 * r_k.abs().max().item()


@pytest.mark.parametrize("R,B,D,masked", [(31680, 64, 192, True), (15872, 64, 256, False), (3968, 64, 384, True),
                                          (997, 1, 1024, True), (130, 5, 68, False)])
def test_norm_bypass_backward_16_byte_form_full_size(dev, R, B, D, masked):
    """s2t_norm_bypass_bwd at the C3 sizes (one, two and four 16-byte chunks per lane; rows that do not
    fill the last trip of four) against autograd through the closed form of BiasNorm + bypass + feature
    mask in fp64 (model/layer/scaling.py:412-476, zipformer.py:1523-1555, 1330-1337)."""
    from speech2text_amd import _native as Nt
    L, st = Nt.lib(), Nt.stream()
    g = torch.Generator().manual_seed(R + D)
    R = (R // B) * B
    x = (torch.randn(R, D, generator=g) * 2).to(dev)
    orig = torch.randn(R, D, generator=g).to(dev)
    bias = (torch.randn(D, generator=g) * 0.1).to(dev)
    ls = torch.tensor([0.3], device=dev)
    k = torch.rand(D, generator=g).to(dev)
    fm = (torch.rand(B, D, generator=g) > 0.3).float().to(dev) if masked else None
    gy = torch.randn(R, D, generator=g).to(dev)
    out, sc = torch.empty(R, D, device=dev), torch.empty(R, device=dev)
    assert L.s2t_norm_bypass_fwd(Nt.fp(x), Nt.fp(bias), Nt.fp(ls), Nt.fp(orig), Nt.fp(k), Nt.fp(fm), B, R, D,
                                 Nt.fp(out), Nt.fp(sc), st) == 0
    dx, d0 = torch.empty(R, D, device=dev), torch.empty(R, D, device=dev)
    dk, db, dl = torch.zeros(D, device=dev), torch.zeros(D, device=dev), torch.zeros(1, device=dev)
    assert L.s2t_norm_bypass_bwd(Nt.fp(x), Nt.fp(bias), Nt.fp(sc), Nt.fp(orig), Nt.fp(k), Nt.fp(gy), Nt.fp(fm), B,
                                 R, D, Nt.fp(dx), Nt.fp(d0), Nt.fp(dk), Nt.fp(db), Nt.fp(dl), st) == 0
    xr, orr, br, lr, kr = (t.double().requires_grad_(True) for t in (x, orig, bias, ls, k))
    scale = ((xr - br) ** 2).mean(dim=1, keepdim=True) ** -0.5 * lr.exp()
    y = orr + (xr * scale - orr) * kr
    if masked:
        y = y * fm.double().repeat(R // B, 1)
    assert (out.double() - y.detach()).abs().max().item() <= 2e-6 * y.detach().abs().max().item()
    y.backward(gy.double())
    for name, got, ref, tol in (("dx", dx, xr.grad, 5e-6), ("d_orig", d0, orr.grad, 2e-6), ("d_scale", dk, kr.grad, 5e-5),
                                ("d_bias", db, br.grad, 5e-5), ("d_log_scale", dl, lr.grad, 5e-5)):
        err = (got.double() - ref).abs().max().item()
        assert err <= tol * max(1.0, ref.abs().max().item()), (name, err)


def test_frontend_output_linear_on_permuted_columns(dev):
    """zk.linear_col_perm (Conv2dSubsampling.out on the channel-last map, reference
    model/layer/subsampling.py:312-319): output, data gradient and the parameter gradients -- returned
    as tensors for parameters outside a flat store -- against F.linear on the permuted weight, fp64."""
    from speech2text_amd import zip_kernels as zk
    g = torch.Generator().manual_seed(5)
    b, t, f, c, n = 3, 50, 19, 128, 192
    x = torch.randn(b, t, f * c, generator=g)
    w = torch.randn(n, c * f, generator=g) * 0.05
    bias = torch.randn(n, generator=g)
    wy = torch.randn(b, t, n, generator=g)
    xr, wr, br = (v.double().requires_grad_(True) for v in (x, w, bias))
    yr = torch.nn.functional.linear(xr, wr.view(n, c, f).permute(0, 2, 1).reshape(n, f * c), br)
    (yr * wy.double()).sum().backward()
    xg, wg, bg = (v.to(dev).requires_grad_(True) for v in (x, w, bias))
    y = zk.linear_col_perm(xg, wg, bg, f, c)
    (y * wy.to(dev)).sum().backward()
    torch.cuda.synchronize()
    for got, ref, tol in ((y.detach(), yr.detach(), 1e-4), (xg.grad, xr.grad, 1e-4), (wg.grad, wr.grad, 2e-4),
                          (bg.grad, br.grad, 2e-4)):
        assert (got.cpu().double() - ref).abs().max().item() <= tol * ref.abs().max().item()
