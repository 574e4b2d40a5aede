"""GPU: the one-node-per-layer executor (speech2text_amd/zip_layer.py) against the
module-by-module layer path on the same weights, inputs and random stream -- with every Balancer
and Whiten forced to fire, with the default probabilities, full and chunk-causal attention.
(The module path itself is pinned to the reference goldens / the oracle in test_gpu_zipformer.py,
which also runs the executor against the oracle.)"""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = dict(feature_dim=80, downsampling_factor=(1, 2, 4), num_encoder_layers=(2, 1, 1),
           feedforward_dim=(128, 192, 192), encoder_dim=(64, 96, 96), encoder_unmasked_dim=(48, 64, 64),
           num_heads=(4, 4, 8), query_head_dim=(16,), value_head_dim=(8,), pos_head_dim=(4,),
           pos_dim=24, cnn_module_kernel=(15, 7, 7), causal=True)


def _build(dev, chunk, left, big=False):
    from speech2text_amd import flat
    from speech2text_amd.model.encoder.zipformer import Zipformer2, Zipformer2Config
    torch.manual_seed(5)
    m = Zipformer2(Zipformer2Config(**CFG, chunk_size=chunk, left_context_frames=left)).to(dev)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("bypass_scale"):
                p.uniform_(0.2, 1.1)                       # some outside [min, max]: limit flips
            elif n.endswith("chunkwise_conv_scale"):
                p.normal_(0, 0.3)
            elif "out_proj" in n or "linear_pos" in n:
                p.mul_(6.0)
            elif big and "self_attn_weights.in_proj" in n:
                p.mul_(5.0)                                # raw scores beyond the penalty limit
    for mod in m.modules():
        if mod.__class__.__name__ == "CompactRelPositionalEncoding":
            mod.dropout.p = 0.0
    store = flat.FlatStore(list(m.parameters()))
    m.train()
    return m, store


def _force(m, on):
    from speech2text_amd.model.layer.scaling import Balancer, Whiten
    for mod in m.modules():
        if isinstance(mod, Balancer) and on:
            mod._modules.pop("prob", None)             # a ScheduledFloat child by default
            mod.prob = 1.0
        if isinstance(mod, Whiten):
            if on:
                mod.min_prob = mod.max_prob = 1.0
            mod.prob = mod.max_prob


def _step(m, store, x, lens, wts, seed, executor):
    from speech2text_amd import zip_layer
    zip_layer.ENABLED = executor
    store.flat_g.zero_()
    random.seed(seed)
    torch.manual_seed(seed)
    xg = x.clone().requires_grad_(True)
    y, _ = m(xg, lens)
    (y * wts).sum().backward()
    torch.cuda.synchronize()
    zip_layer.ENABLED = True
    return y.detach().clone(), xg.grad.clone(), store.flat_g.clone()


@pytest.mark.parametrize("force,big", [(True, False), (False, False), (False, True)])
@pytest.mark.parametrize("chunk,left", [((-1,), (-1,)), ((8,), (16,))])
def test_executor_matches_module_path(dev, monkeypatch, force, big, chunk, left):
    """big: attention projections scaled so raw scores exceed penalize_abs_values_gt's limit -- on
    the ~10 % of calls that draw the penalty the executor must take its materialised branch."""
    from speech2text_amd import rng, zip_layer
    monkeypatch.setattr(rng, "rand", lambda *s, device=None, dtype=torch.float32:
                        torch.rand(*s, dtype=dtype).to(device))
    m, store = _build(dev, chunk, left, big)
    active0 = zip_layer.STATS["penalty_active"]
    _force(m, force)
    g = torch.Generator().manual_seed(9)
    B, T = 4, 211
    x = (torch.randn(B, T, 80, generator=g) * 2).to(dev)
    lens = torch.tensor([211, 190, 97, 64]).to(dev)
    with torch.no_grad():
        wts = torch.randn(m(x, lens)[0].shape, generator=g).to(dev)
    served = 0
    for seed in range(14 if big else 6):
        _force(m, force)
        c0 = zip_layer.CALLS[0]
        y1, gx1, gp1 = _step(m, store, x, lens, wts, seed, True)
        served += zip_layer.CALLS[0] - c0
        _force(m, force)                      # Whiten.prob moves in backward: same start state
        y0, gx0, gp0 = _step(m, store, x, lens, wts, seed, False)
        torch.testing.assert_close(y1, y0, atol=2e-5, rtol=1e-4)
        torch.testing.assert_close(gx1, gx0, atol=2e-5, rtol=2e-3)
        scale = gp0.abs().max()
        assert (gp1 - gp0).abs().max() / scale < 2e-4, (seed, (gp1 - gp0).abs().max(), scale)
        # per parameter, so a small tensor's gradient cannot hide behind a large one's scale
        for p, (o, n) in zip(store.params, zip(store.offsets, store.lengths)):
            a, b = gp1[o:o + n], gp0[o:o + n]
            assert (a - b).abs().max() <= 2e-3 * b.abs().max() + 1e-6, (seed, tuple(p.shape))
    assert served == 4 * (14 if big else 6)   # every layer call, penalty draws included
    if big:
        assert zip_layer.STATS["penalty_active"] > active0


@pytest.mark.parametrize("chunk,left", [((-1,), (-1,)), ((8,), (16,))])
@pytest.mark.parametrize("rv", [0.0, 0.2, 0.5])
def test_native_executor_matches_python_executor(dev, monkeypatch, rv, chunk, left):
    """csrc/zip_layer.hip (one C call per layer pass) against zip_layer._LayerFn, the same launch
    sequence issued from Python, with every random draw pinned to rv: 0.0 fires every Balancer /
    Whiten / limit / the score penalty, 0.2 the Balancers and limits, 0.5 the limits only.  The
    forward output is bit-identical; so is the input gradient when no statistics kernel runs
    (rv 0.5) -- the Balancer / Whiten statistics and the weight gradients are summed with fp32
    atomics, so two runs of ONE path already differ in the last bits there."""
    from speech2text_amd import rng, zip_layer, zip_native
    from speech2text_amd.model.layer import scaling as S
    monkeypatch.setattr(rng, "rand", lambda *s, device=None, dtype=torch.float32:
                        torch.rand(*s, dtype=dtype).to(device))
    monkeypatch.setattr(S, "_rand", lambda: rv)
    m, store = _build(dev, chunk, left)
    g = torch.Generator().manual_seed(11)
    B, T = 4, 203
    x = (torch.randn(B, T, 80, generator=g) * 2).to(dev)
    lens = torch.tensor([203, 180, 97, 64]).to(dev)
    with torch.no_grad():
        wts = torch.randn(m(x, lens)[0].shape, generator=g).to(dev)

    def step(native):
        monkeypatch.setattr(zip_native, "ENABLED", native)
        _force(m, False)
        return _step(m, store, x, lens, wts, 3, True)

    step(False)                                   # times the GEMM shape buckets (Python executor)
    n0 = list(zip_native.CALLS)
    y1, gx1, gp1 = step(True)
    assert zip_native.CALLS[0] - n0[0] == 4 and zip_native.CALLS[1] - n0[1] == 4    # every layer call
    y0, gx0, gp0 = step(False)
    assert zip_native.CALLS[0] - n0[0] == 4
    assert torch.equal(y1, y0)
    if rv == 0.5:
        assert torch.equal(gx1, gx0)
    else:
        torch.testing.assert_close(gx1, gx0, atol=2e-5, rtol=2e-3)
    for p, (o, n) in zip(store.params, zip(store.offsets, store.lengths)):
        a, b = gp1[o:o + n], gp0[o:o + n]
        assert (a - b).abs().max() <= 2e-3 * b.abs().max() + 1e-6, tuple(p.shape)
    # the Whiten modules' probabilities moved the same way in both backward passes
    from speech2text_amd.model.layer.scaling import Whiten
    probs1 = None
    for native in (True, False):
        step(native)
        probs = [w.prob for w in m.modules() if isinstance(w, Whiten)]
        assert probs1 is None or probs == probs1
        probs1 = probs


def test_native_executor_sees_an_in_place_edit_of_any_weight(dev, monkeypatch):
    """The native executor's descriptor holds the addresses of the weights' pre-split bf16 pieces
    (planes.py).  An in-place torch edit of ONE matrix of a layer -- not the first the freshness
    check looks at, no optimizer step, no epoch bump (a partial load_state_dict, a manual
    weight.mul_) -- must be seen through that parameter's _version: the forward after the edit is
    bit-identical to the Python executor's, which looks every weight up."""
    from speech2text_amd import rng, zip_layer, zip_native
    from speech2text_amd.model.layer import scaling as S
    monkeypatch.setattr(rng, "rand", lambda *s, device=None, dtype=torch.float32:
                        torch.rand(*s, dtype=dtype).to(device))
    monkeypatch.setattr(S, "_rand", lambda: 0.5)
    m, store = _build(dev, (-1,), (-1,))
    g = torch.Generator().manual_seed(11)
    B, T = 4, 203
    x = (torch.randn(B, T, 80, generator=g) * 2).to(dev)
    lens = torch.tensor([203, 180, 97, 64]).to(dev)
    with torch.no_grad():
        wts = torch.randn(m(x, lens)[0].shape, generator=g).to(dev)

    def step(native):
        monkeypatch.setattr(zip_native, "ENABLED", native)
        _force(m, False)
        return _step(m, store, x, lens, wts, 3, True)

    step(False)
    y_before, _, _ = step(True)
    layer = [mod for mod in m.modules() if mod.__class__.__name__ == "Zipformer2EncoderLayer"][-1]
    with torch.no_grad():
        layer.feed_forward3.out_proj.weight.mul_(1.5)          # (a late weight of the last layer)
    n0 = list(zip_native.CALLS)
    y1, gx1, _ = step(True)
    assert zip_native.CALLS[0] - n0[0] == 4
    y0, gx0, _ = step(False)
    assert not torch.equal(y1, y_before)
    assert torch.equal(y1, y0) and torch.equal(gx1, gx0)
