"""GPU parity: fbank / CTC / RNN-T lattice kernels (through the C ABI) vs oracle + goldens."""
import os

import numpy as np
import pytest
import torch

from oracle import ctc as octc
from oracle import fbank as ofbank
from oracle import k2_rnnt as K

pytestmark = pytest.mark.gpu


def _kern():
    from speech2text_amd import kernels
    return kernels


# ------------------------------------------------------------------ fbank
def test_fbank_vs_reference_archive(golden_dir, dev):
    k = _kern()
    g = np.load(os.path.join(golden_dir, "fbank_script64.npz"))
    tab = k.FbankTables(num_mel_bins=64, device=dev)
    pcms = [g[f"pcm{i}"] for i in range(6)]
    nmax = max(p.shape[0] for p in pcms)
    batch = np.zeros((len(pcms), nmax), np.float32)
    for i, p in enumerate(pcms):
        batch[i, :p.shape[0]] = p
    lens = torch.tensor([p.shape[0] for p in pcms], dtype=torch.int64, device=dev)
    feats, frames = k.fbank_batch(torch.from_numpy(batch).to(dev), lens, tab)
    feats = feats.cpu().numpy()
    frames = frames.cpu().numpy()
    for i in range(6):
        ref = g[f"feat{i}"]
        assert frames[i] == ref.shape[0]
        # tolerance: fp32 FFT of a different factorisation; log amplifies near-eps bins
        np.testing.assert_allclose(feats[i, :ref.shape[0]], ref, rtol=0, atol=3e-3)
        assert np.abs(feats[i, :ref.shape[0]] - ref).mean() < 5e-5
        assert (feats[i, ref.shape[0]:] == 0).all()          # batch padding value 0


@pytest.mark.parametrize("bins,high", [(80, 0.0), (80, -400.0), (23, 0.0)])
def test_fbank_vs_oracle_ragged_with_cmvn(dev, bins, high):
    k = _kern()
    rng = np.random.default_rng(5)
    lens = [16000, 399, 400, 7777, 48000, 160 * 33 + 400]
    batch = np.zeros((len(lens), max(lens)), np.float32)
    for i, n in enumerate(lens):
        batch[i, :n] = np.clip(0.1 * rng.standard_normal(n) + 0.2 * np.sin(np.arange(n) * 0.05 * (i + 1)), -1, 1)
    tab = k.FbankTables(num_mel_bins=bins, high_freq=high, device=dev)
    mean = torch.linspace(-1, 1, bins, device=dev)
    istd = torch.linspace(0.5, 2, bins, device=dev)
    feats, frames = k.fbank_batch(torch.from_numpy(batch).to(dev),
                                  torch.tensor(lens, device=dev), tab, mean, istd)
    feats = feats.cpu().numpy()
    for i, n in enumerate(lens):
        ref = ofbank.fbank(batch[i, :n], bins, high_freq=high)
        assert frames[i].item() == ref.shape[0]
        ref = (ref - mean.cpu().numpy()) * istd.cpu().numpy()
        np.testing.assert_allclose(feats[i, :ref.shape[0]], ref, rtol=0, atol=6e-3)
        pad = (0 - mean.cpu().numpy()) * istd.cpu().numpy()
        if ref.shape[0] < feats.shape[1]:
            np.testing.assert_allclose(feats[i, ref.shape[0]:], np.broadcast_to(pad, feats[i, ref.shape[0]:].shape), atol=1e-6)


def test_fbank_empty_batch(dev):
    k = _kern()
    tab = k.FbankTables(80, device=dev)
    f, n = k.fbank_batch(torch.zeros((2, 100), device=dev), torch.tensor([100, 50], device=dev), tab)
    assert f.shape == (2, 0, 80) and (n == 0).all()


# ------------------------------------------------------------------ CTC
@pytest.mark.parametrize("ci", [0, 1, 2, 3])
def test_ctc_vs_reference_golden(golden_dir, dev, ci):
    k = _kern()
    g = np.load(os.path.join(golden_dir, "ctc_ref.npz"))
    logits = torch.from_numpy(g[f"logits{ci}"]).to(dev).requires_grad_(True)
    loss = k.ctc_loss(logits, torch.from_numpy(g[f"targets{ci}"]).to(dev),
                      torch.from_numpy(g[f"in_len{ci}"]).to(dev),
                      torch.from_numpy(g[f"tgt_len{ci}"]).to(dev))
    loss.backward()
    # fp32 tolerance (north_star: loss within 1e-3 relative; we hold 1e-5)
    np.testing.assert_allclose(loss.item(), g[f"loss{ci}"], rtol=1e-5)
    np.testing.assert_allclose(logits.grad.cpu().numpy(), g[f"grad{ci}"], atol=2e-5, rtol=1e-3)


def test_ctc_vs_oracle_c2_shape_and_edges(dev):
    k = _kern()
    rng = np.random.default_rng(11)
    B, T, V, U = 8, 249, 128, 40
    logits = (rng.standard_normal((B, T, V)) * 3).astype(np.float32)
    tl = rng.integers(0, U + 1, size=B); tl[0] = U; tl[1] = 0       # empty target
    il = rng.integers(100, T + 1, size=B); il[0] = T; il[2] = 1; tl[2] = 1
    tg = rng.integers(1, V, size=(B, U))
    tg[3, :10] = 7                                                  # long repeat run
    ref_loss, ref_grad, _ = octc.ctc_loss(logits, tg, il, tl)
    lg = torch.from_numpy(logits).to(dev).requires_grad_(True)
    for red in ["mean", "sum"]:
        lg.grad = None
        loss = k.ctc_loss(lg, torch.from_numpy(tg).to(dev), torch.from_numpy(il).to(dev),
                          torch.from_numpy(tl).to(dev), reduction=red)
        loss.backward()
        rl, rg, _ = octc.ctc_loss(logits, tg, il, tl, reduction=red)
        np.testing.assert_allclose(loss.item(), rl, rtol=2e-5)
        # nll here is ~700 nats over 249 frames: fp32 log-domain alpha/beta carry ~1e-3 of
        # accumulated rounding (the reference's fp32 nn.CTCLoss does too); oracle is fp64
        np.testing.assert_allclose(lg.grad.cpu().numpy(), rg, atol=3e-3 if red == "sum" else 3e-5,
                                   rtol=2e-3)


# ------------------------------------------------------------------ RNN-T
def _case(seed, B, T, S, C, dev):
    g = torch.Generator().manual_seed(seed)
    am = torch.randn(B, T, C, generator=g) * 2
    lm = torch.randn(B, S + 1, C, generator=g) * 2
    sym = torch.randint(1, C, (B, S), generator=g)
    tl = torch.randint(max(1, S // 2), S + 1, (B,), generator=g); tl[0] = S
    el = torch.randint(max(S + 1, T // 2), T + 1, (B,), generator=g); el[0] = T
    return am, lm, sym, tl, el


@pytest.mark.parametrize("B,T,S,C", [(3, 12, 5, 9), (4, 70, 30, 33), (2, 130, 90, 17)])
def test_mutual_information_vs_oracle(dev, B, T, S, C):
    k = _kern()
    am, lm, sym, tl, el = _case(3, B, T, S, C, dev)
    bnd = torch.zeros(B, 4, dtype=torch.int64); bnd[:, 2] = tl; bnd[:, 3] = el
    px, py = K.get_rnnt_logprobs_smoothed(lm, am, sym, 0, bnd)
    p, ans, gx, gy = K.mutual_information_np(px.numpy(), py.numpy(), bnd.numpy())
    a2, p2, gx2, gy2 = k.mutual_information(px.to(dev).contiguous(), py.to(dev).contiguous(), bnd.to(dev))
    np.testing.assert_allclose(a2.cpu().numpy(), ans, rtol=2e-5, atol=1e-4)
    np.testing.assert_allclose(gx2.cpu().numpy(), gx, atol=2e-4)
    np.testing.assert_allclose(gy2.cpu().numpy(), gy, atol=2e-4)


@pytest.mark.parametrize("B,T,S,C,R", [(3, 12, 5, 9, 3), (4, 70, 30, 33, 5), (2, 40, 12, 500, 5)])
def test_simple_loss_ranges_pruned_loss_vs_oracle(dev, B, T, S, C, R):
    k = _kern()
    am, lm, sym, tl, el = _case(4, B, T, S, C, dev)
    am_c = am.clone().requires_grad_(True); lm_c = lm.clone().requires_grad_(True)
    logits, bnd, ranges, simple = K.joiner_pruned(am_c, lm_c, sym, tl, el, R)
    pruned = K.rnnt_loss_pruned(logits, sym, ranges, 0, bnd)
    (0.5 * simple + 0.5 * pruned).backward()

    am_g = am.to(dev).requires_grad_(True); lm_g = lm.to(dev).requires_grad_(True)
    bnd_g = k.make_boundary(tl, el, dev)
    neg, gx, gy = k.rnnt_simple_loss(lm_g, am_g, sym.to(dev), bnd_g)
    np.testing.assert_allclose(neg.mean().item(), simple.item(), rtol=1e-4)
    rg = k.rnnt_prune_ranges(gx, gy, bnd_g, R)
    assert torch.equal(rg.cpu(), ranges)                       # integer work: bit-exact
    pl = k.rnnt_pruned_joiner_loss(am_g, lm_g, rg, sym.to(dev), bnd_g)
    np.testing.assert_allclose(pl.mean().item(), pruned.item(), rtol=1e-4)
    (0.5 * neg.mean() + 0.5 * pl.mean()).backward()
    np.testing.assert_allclose(am_g.grad.cpu().numpy(), am_c.grad.numpy(), atol=3e-5, rtol=2e-3)
    np.testing.assert_allclose(lm_g.grad.cpu().numpy(), lm_c.grad.numpy(), atol=3e-5, rtol=2e-3)
    # materialised-lattice entry point gives the same loss and d(logits)
    lat = logits.detach().to(dev).requires_grad_(True)
    ll = k.rnnt_lattice_loss(lat, rg, sym.to(dev), bnd_g)
    np.testing.assert_allclose(ll.mean().item(), pruned.item(), rtol=1e-4)


def test_full_rnnt_loss_vs_oracle(dev):
    k = _kern()
    torch.manual_seed(5)
    B, T, U, V = 3, 25, 9, 31
    logits = torch.randn(B, T, U + 1, V) * 2
    tg = torch.randint(1, V, (B, U))
    tl = torch.tensor([U, 4, 1]); el = torch.tensor([T, 20, 11])
    lc = logits.clone().requires_grad_(True)
    ref = K.rnnt_loss_full(lc, tg, el, tl)
    ref.backward()
    lg = logits.to(dev).requires_grad_(True)
    out = k.rnnt_lattice_loss(lg, None, tg.to(dev), k.make_boundary(tl, el, dev)).mean()
    out.backward()
    np.testing.assert_allclose(out.item(), ref.item(), rtol=1e-5)
    np.testing.assert_allclose(lg.grad.cpu().numpy(), lc.grad.numpy(), atol=2e-5, rtol=2e-3)


# ------------------------------------------------------------------ BEST-RQ
def test_bestrq_labels_bit_exact_and_masks(golden_dir, dev):
    from oracle import best_rq as obr
    from speech2text_amd.model.ssl.best_rq import (BestRQLayer, BestRQLayerConfig,
                                                   MaskingStrategyConfig)
    g = np.load(os.path.join(golden_dir, "bestrq_ref.npz"))
    for ci, (K, D, ncb) in enumerate([(256, 16, 1), (64, 8, 2), (8192, 16, 1)]):
        basis = str(g[f"basis{ci}"])
        layer = BestRQLayer(BestRQLayerConfig(feat_dim=80, num_codebooks=ncb, codebook_dim=D,
                                              codebook_size=K, label_basis=basis),
                            MaskingStrategyConfig(mask_proportion=0.5, mean_span_length=1,
                                                  span_select_type="static", seed=5)).to(dev)
        with torch.no_grad():
            layer._projector.copy_(torch.from_numpy(g[f"projector{ci}"]))
            for j in range(ncb):
                layer._codebooks[j].copy_(torch.from_numpy(g[f"codebook{ci}_{j}"]))
        raw = torch.from_numpy(g[f"raw{ci}"]).to(dev)
        aug = torch.from_numpy(g[f"aug{ci}"]).to(dev)
        np.random.seed(99 + ci)                       # the reference's global np.random.rand() draw
        out = layer(raw, aug.clone(), torch.from_numpy(g[f"length{ci}"]))
        lab = out["labels"].cpu().numpy()
        ora = obr.make_labels(g[f"raw{ci}"], g[f"projector{ci}"],
                              [g[f"codebook{ci}_{j}"] for j in range(ncb)])
        assert (lab == ora).all()                     # bit-exact vs oracle
        assert (lab == g[f"labels{ci}"]).all()        # and vs the reference's own labels
        assert (out["masked_dim"].cpu().numpy() == g[f"masked_dim{ci}"]).all()
        changed = (out["masked_feats"] != aug).any(-1).cpu().numpy()
        assert (changed == g[f"changed{ci}"]).all()   # same frames overwritten with noise


def test_bestrq_labels_c5_shape(dev):
    from oracle import best_rq as obr
    from speech2text_amd.model.ssl.best_rq import (BestRQLayer, BestRQLayerConfig,
                                                   MaskingStrategyConfig)
    torch.manual_seed(3)
    layer = BestRQLayer(BestRQLayerConfig(codebook_dim=16, codebook_size=8192, label_basis="cosine"),
                        MaskingStrategyConfig(mask_proportion=0.5)).to(dev)
    feats = torch.randn(2, 2998, 80) * 3
    lab = layer.make_labels(feats.to(dev)).cpu().numpy()
    assert lab.shape == (1, 2, 748)
    ora = obr.make_labels(feats.numpy(), layer._projector.cpu().numpy(), [layer._codebooks[0].cpu().numpy()])
    assert (lab == ora).all()
