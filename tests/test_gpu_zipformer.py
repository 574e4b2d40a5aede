"""GPU parity: the product Zipformer2 (HIP kernels + rocBLAS GEMMs) against (a) goldens captured
from the reference and (b) the oracle at a larger ragged configuration."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import zipformer as Z

pytestmark = pytest.mark.gpu

TINY = dict(feature_dim=80, downsampling_factor=(1, 2, 4), num_encoder_layers=(1, 1, 1),
            feedforward_dim=(64, 96, 96), encoder_dim=(32, 48, 48), encoder_unmasked_dim=(24, 32, 32),
            num_heads=(4, 4, 4), query_head_dim=(8,), value_head_dim=(4,), pos_head_dim=(4,),
            pos_dim=16, cnn_module_kernel=(7, 5, 5), causal=True)
MID = dict(feature_dim=80, downsampling_factor=(1, 2, 4), num_encoder_layers=(2, 1, 1),
           feedforward_dim=(128, 192, 192), encoder_dim=(64, 96, 96), encoder_unmasked_dim=(48, 64, 64),
           num_heads=(4, 4, 8), query_head_dim=(16,), value_head_dim=(8,), pos_head_dim=(4,),
           pos_dim=24, cnn_module_kernel=(15, 7, 7), causal=True)


def _zcfg(c):
    n = len(c["downsampling_factor"])
    t = lambda v: tuple(v) * (n if len(v) == 1 else 1)   # noqa: E731
    return dict(downsampling_factor=c["downsampling_factor"], num_encoder_layers=c["num_encoder_layers"],
                encoder_dim=c["encoder_dim"], encoder_unmasked_dim=c["encoder_unmasked_dim"],
                num_heads=c["num_heads"], query_head_dim=t(c["query_head_dim"]),
                pos_head_dim=t(c["pos_head_dim"]), cnn_module_kernel=c["cnn_module_kernel"],
                pos_dim=c["pos_dim"])


def _model(cfg, chunk, left, dev):
    from speech2text_amd.model.encoder.zipformer import Zipformer2, Zipformer2Config
    return Zipformer2(Zipformer2Config(**cfg, chunk_size=chunk, left_context_frames=left)).to(dev)


@pytest.fixture
def cpu_rng(monkeypatch):
    from speech2text_amd import rng
    monkeypatch.setattr(rng, "rand", lambda *s, device=None, dtype=torch.float32:
                        torch.rand(*s, dtype=dtype).to(device))


def _train_step(m, x, lens, wts, rv=0.0):
    real = random.random
    random.random = lambda: rv           # 0.0: every Balancer / Whiten / limit / penalty fires
    try:
        torch.manual_seed(7)
        y, yl = m(x, lens)
        loss = (y * wts).sum()
        loss.backward()
    finally:
        random.random = real
    return y, loss


@pytest.mark.parametrize("tag,chunk,left", [("full", (-1,), (-1,)), ("chunk8", (8,), (16,))])
def test_tiny_vs_reference_goldens(golden_dir, dev, cpu_rng, tag, chunk, left):
    g = np.load(os.path.join(golden_dir, f"zipformer_tiny_{tag}.npz"))
    m = _model(TINY, chunk, left, dev)
    m.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")})
    x = torch.from_numpy(g["x"]).to(dev)
    lens = torch.from_numpy(g["lens"]).to(dev)
    m.eval()
    with torch.no_grad():
        y, yl = m(x, lens)
    assert (yl.cpu().numpy() == g["eval_lens"]).all()
    # fp32 tolerance for encoder activations (north_star: "within a stated fp32 tolerance")
    np.testing.assert_allclose(y.cpu().numpy(), g["eval_out"], atol=5e-5, rtol=1e-3)
    # deterministic training step: the golden used torch's CPU dropout on pos_emb (p=0.15);
    # dropout masks cannot be reproduced across devices, so compare with dropout disabled
    # against the oracle (itself pinned to the golden WITH dropout on CPU)
    m.train()
    for mod in m.modules():
        if mod.__class__.__name__ == "CompactRelPositionalEncoding":
            mod.dropout.p = 0.0
    sd = {k[3:]: torch.from_numpy(g[k]).clone().requires_grad_(True) for k in g.files if k.startswith("sd.")}
    xc = torch.from_numpy(g["x"]).requires_grad_(True)
    torch.manual_seed(7)
    yo, _ = Z.zipformer_forward(sd, _zcfg(TINY), xc, torch.from_numpy(g["lens"]),
                                Z.Ctl(True, lambda: 0.0, pos_dropout=0.0), chunk[0],
                                -1 if left[0] < 0 else max(1, left[0] // chunk[0]))
    wts = torch.from_numpy(g["train_wts"])
    (yo * wts).sum().backward()
    xg = x.clone().requires_grad_(True)
    y, loss = _train_step(m, xg, lens, wts.to(dev))
    np.testing.assert_allclose(y.detach().cpu().numpy(), yo.detach().numpy(), atol=5e-5, rtol=1e-3)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xc.grad.numpy(), atol=2e-5, rtol=5e-3)
    for n, p in m.named_parameters():
        ref = sd[n].grad.numpy() if sd[n].grad is not None else np.zeros(p.shape, np.float32)
        got = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(ref)
        denom = np.abs(ref).max() + 1e-6
        assert np.abs(got - ref).max() / denom < 5e-3, (n, np.abs(got - ref).max(), denom)


@pytest.mark.parametrize("rv,store", [(0.0, False), (0.5, False), (0.5, True), (0.2, True),
                                      (0.0, True)])
def test_mid_ragged_vs_oracle(dev, cpu_rng, rv, store):
    """rv = value returned by random.random(): 0.0 -> all gradient shaping + the score penalty
    (materialised attention path); 0.5 -> no Balancer/Whiten/penalty, limit_param_value only,
    which exercises the fused attention path with deferred (never materialised) dW; 0.2 -> the
    default-probability Balancers, every Whiten and limit_param_value, no penalty.
    store: parameters in a FlatStore (as under the Trainer) -> the one-node-per-layer executor
    (speech2text_amd/zip_layer.py) serves every layer call (the score penalty through its
    flag-and-fallback branch at rv = 0.0); without a store the module-by-module path runs."""
    from speech2text_amd import flat, zip_layer
    torch.manual_seed(11)
    m = _model(MID, (-1,), (-1,), dev)
    if store:
        flat.FlatStore(list(m.parameters()))
    calls0 = zip_layer.CALLS[0]
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("bypass_scale"):
                p.uniform_(0.2, 0.9)
            elif n.endswith("chunkwise_conv_scale"):
                p.normal_(0, 0.3)
            elif "out_proj" in n or "pointwise_conv2" in n or "linear_pos" in n:
                p.mul_(6.0)
    g = torch.Generator().manual_seed(3)
    B, T = 5, 263
    x = torch.randn(B, T, 80, generator=g) * 2
    lens = torch.tensor([263, 250, 200, 131, 77])
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    m.eval()
    with torch.no_grad():
        y, yl = m(x.to(dev), lens.to(dev))
        yo, ylo = Z.zipformer_forward(sd, _zcfg(MID), x, lens, Z.Ctl(False))
    assert torch.equal(yl.cpu(), ylo)
    np.testing.assert_allclose(y.cpu().numpy(), yo.numpy(), atol=1e-4, rtol=2e-3)
    m.train()
    for mod in m.modules():
        if mod.__class__.__name__ == "CompactRelPositionalEncoding":
            mod.dropout.p = 0.0
    wts = torch.randn(yo.shape, generator=g)
    xc = x.clone().requires_grad_(True)
    torch.manual_seed(7)
    yo, _ = Z.zipformer_forward(sd, _zcfg(MID), xc, lens, Z.Ctl(True, lambda: rv, pos_dropout=0.0))
    (yo * wts).sum().backward()
    xg = x.to(dev).requires_grad_(True)
    y, _ = _train_step(m, xg, lens.to(dev), wts.to(dev), rv)
    n_layers = sum(MID["num_encoder_layers"])
    assert zip_layer.CALLS[0] - calls0 == (n_layers if store else 0)
    np.testing.assert_allclose(y.detach().cpu().numpy(), yo.detach().numpy(), atol=1e-4, rtol=2e-3)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xc.grad.numpy(), atol=5e-5, rtol=1e-2)
    worst = 0.0
    for n, p in m.named_parameters():
        ref = sd[n].grad.numpy() if sd[n].grad is not None else np.zeros(p.shape, np.float32)
        got = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(ref)
        worst = max(worst, np.abs(got - ref).max() / (np.abs(ref).max() + 1e-6))
    assert worst < 1e-2, worst


# ------------------------------------------------------------------ streaming
def _stream_model(golden_dir, dev):
    g = np.load(os.path.join(golden_dir, "zipformer_tiny_stream.npz"))
    chunk, left = int(g["chunk"]), int(g["left"])
    from speech2text_amd.model.encoder.zipformer import Zipformer2, Zipformer2Config
    m = Zipformer2(Zipformer2Config(**TINY, chunk_size=(chunk,), left_context_frames=(left,),
                                    for_ctc=True, num_tokens=13))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    return g, m.to(dev).eval(), chunk, left


def test_streaming_step_vs_reference_goldens(golden_dir, dev):
    """Zipformer2.get_init_states / streaming_step (HIP kernels) against the reference's own
    streaming_step: 6 consecutive chunks, encoder output, CTC scores and every carried state."""
    g, m, chunk, left = _stream_model(golden_dir, dev)
    feats = torch.from_numpy(g["feats"]).to(dev)
    B, T = feats.shape[0], 2 * chunk + 13
    st = m.get_init_states(B, dev)
    assert len(st) == int(g["n_states"])
    for i, s in enumerate(st):
        assert tuple(s.shape) == tuple(g[f"init_shape.{i}"]), i
    for c in range(6):
        x = feats[:, 2 * chunk * c:2 * chunk * c + T]
        m._for_ctc = False
        raw, _ = m.streaming_step(x, st)
        m._for_ctc = True
        y, st = m.streaming_step(x, st)
        np.testing.assert_allclose(raw.cpu().numpy(), g[f"raw.{c}"], atol=5e-5, rtol=2e-4)
        np.testing.assert_allclose(y.cpu().numpy(), g[f"out.{c}"], atol=5e-5, rtol=2e-4)
        if c in (0, 2):
            for i, s in enumerate(st):
                assert tuple(s.shape) == tuple(g[f"state{c}.{i}"].shape), i
                np.testing.assert_allclose(s.cpu().numpy(), g[f"state{c}.{i}"], atol=5e-5,
                                           rtol=2e-4, err_msg=f"state {i} after chunk {c}")
    for i, s in enumerate(st):
        np.testing.assert_allclose(s.cpu().numpy(), g[f"final_state.{i}"], atol=5e-5, rtol=2e-4)
    assert (st[-1].cpu().numpy() == 6 * chunk).all()


def test_streaming_step_vs_oracle_mid(dev):
    """A wider causal model (MID dims, chunk 16 / left 32, batch 3) against the oracle's
    streaming restatement, including the steady state where every cache is full."""
    torch.manual_seed(11)
    chunk, left, B = 16, 32, 3
    m = _model(MID, (chunk,), (left,), "cpu")
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("bypass_scale"):
                p.uniform_(0.2, 0.9)
            elif n.endswith("chunkwise_conv_scale"):
                p.normal_(0, 0.3)
            elif "out_proj" in n or "pointwise_conv2" in n or "linear_pos" in n:
                p.mul_(6.0)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    zc = dict(_zcfg(MID), value_head_dim=(8, 8, 8))
    m = m.to(dev).eval()
    T = 2 * chunk + 13
    feats = torch.randn(B, 2 * chunk * 4 + T, 80) * 2.0
    st_o = Z.streaming_init_states(zc, B, left)
    st = m.get_init_states(B, dev)
    for c in range(5):
        x = feats[:, 2 * chunk * c:2 * chunk * c + T]
        with torch.no_grad():
            yo, st_o = Z.streaming_step(sd, zc, x, st_o, chunk, left)
        y, st = m.streaming_step(x.to(dev), st)
        np.testing.assert_allclose(y.cpu().numpy(), yo.numpy(), atol=1e-4, rtol=5e-4)
    for a, b in zip(st, st_o):
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), atol=1e-4, rtol=5e-4)


def test_streaming_rejects_cpu_and_training(golden_dir, dev):
    g, m, chunk, left = _stream_model(golden_dir, dev)
    x = torch.zeros(1, 2 * chunk + 13, 80)
    with pytest.raises(RuntimeError):
        m.streaming_step(x, m.get_init_states(1))            # CPU tensors: no fallback
    with pytest.raises(ValueError):
        m.streaming_step(x[:, :-1].to(dev), m.get_init_states(1, dev))
    m.train()
    with pytest.raises(RuntimeError):
        m.streaming_step(x.to(dev), m.get_init_states(1, dev))


def test_streaming_session_graph_matches_eager(golden_dir, dev):
    """StreamingSession (one hipGraph per chunk step, states updated inside the graph) is
    bit-identical to eager streaming_step, across reset()."""
    from speech2text_amd.model.encoder.zipformer_streaming import StreamingSession
    g, m, chunk, left = _stream_model(golden_dir, dev)
    feats = torch.from_numpy(g["feats"]).to(dev)
    B, T = feats.shape[0], 2 * chunk + 13
    sess = StreamingSession(m, B, dev)
    for rep in range(2):
        st = m.get_init_states(B, dev)
        sess.reset()
        for c in range(6):
            x = feats[:, 2 * chunk * c:2 * chunk * c + T]
            y, st = m.streaming_step(x, st)
            assert torch.equal(sess.step(x), y), (rep, c)
        for a, b in zip(sess.states, st):
            assert torch.equal(a, b)
    np.testing.assert_allclose(sess.out.cpu().numpy(), g["out.5"], atol=5e-5, rtol=2e-4)
    with pytest.raises(ValueError):
        sess.step(feats[:, :T - 1])
