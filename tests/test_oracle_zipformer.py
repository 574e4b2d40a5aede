"""CPU: oracle/zipformer.py against goldens captured from the reference Zipformer2."""
import os

import numpy as np
import pytest
import torch

from oracle import zipformer as Z

TINY = dict(downsampling_factor=(1, 2, 4), num_encoder_layers=(1, 1, 1), encoder_dim=(32, 48, 48),
            encoder_unmasked_dim=(24, 32, 32), num_heads=(4, 4, 4), query_head_dim=(8, 8, 8),
            pos_head_dim=(4, 4, 4), value_head_dim=(4, 4, 4), cnn_module_kernel=(7, 5, 5),
            pos_dim=16, feedforward_dim=(64, 96, 96))


def load(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, f"zipformer_tiny_{tag}.npz"))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    return g, sd


@pytest.mark.parametrize("tag,chunk,lcc", [("full", -1, -1), ("chunk8", 8, 2)])
def test_eval_forward(golden_dir, tag, chunk, lcc):
    g, sd = load(golden_dir, tag)
    with torch.no_grad():
        y, yl = Z.zipformer_forward(sd, TINY, torch.from_numpy(g["x"]), torch.from_numpy(g["lens"]),
                                    Z.Ctl(training=False), chunk, lcc)
    assert (yl.numpy() == g["eval_lens"]).all()
    np.testing.assert_allclose(y.numpy(), g["eval_out"], atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize("tag,chunk,lcc", [("full", -1, -1), ("chunk8", 8, 2)])
def test_deterministic_train_step_grads(golden_dir, tag, chunk, lcc):
    g, sd = load(golden_dir, tag)
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    torch.manual_seed(7)
    y, _ = Z.zipformer_forward(sd, TINY, x, torch.from_numpy(g["lens"]),
                               Z.Ctl(training=True, rand=lambda: 0.0), chunk, lcc)
    np.testing.assert_allclose(y.detach().numpy(), g["train_out"], atol=2e-5, rtol=1e-4)
    loss = (y * torch.from_numpy(g["train_wts"])).sum()
    loss.backward()
    np.testing.assert_allclose(loss.item(), g["train_loss"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(x.grad.numpy(), g["grad.x"], atol=1e-5, rtol=2e-3)
    worst = 0.0
    for k, v in sd.items():
        ref = g["grad." + k]
        got = v.grad.numpy() if v.grad is not None else np.zeros_like(ref)
        denom = np.abs(ref).max() + 1e-6
        worst = max(worst, np.abs(got - ref).max() / denom)
        np.testing.assert_allclose(got, ref, atol=2e-3 * denom, rtol=0, err_msg=k)
    assert worst < 2e-3


def test_streaming_step_vs_reference(golden_dir):
    """oracle streaming_step against the reference's Zipformer2.streaming_step over 6 chunks from
    get_init_states: per-chunk encoder output, CTC log-softmax output and the carried states."""
    g, sd = load(golden_dir, "stream")
    chunk, left = int(g["chunk"]), int(g["left"])
    feats = torch.from_numpy(g["feats"])
    B, T = feats.shape[0], 2 * chunk + 13
    st = Z.streaming_init_states(TINY, B, left)
    assert len(st) == int(g["n_states"])
    for i, s in enumerate(st):
        assert tuple(s.shape) == tuple(g[f"init_shape.{i}"]), i
    with torch.no_grad():
        for c in range(6):
            x = feats[:, 2 * chunk * c:2 * chunk * c + T]
            raw, _ = Z.streaming_step(sd, TINY, x, st, chunk, left)
            y, st = Z.streaming_step(sd, TINY, x, st, chunk, left, for_ctc=True)
            np.testing.assert_allclose(raw.numpy(), g[f"raw.{c}"], atol=2e-5, rtol=1e-4)
            np.testing.assert_allclose(y.numpy(), g[f"out.{c}"], atol=2e-5, rtol=1e-4)
            if c in (0, 2):
                for i, s in enumerate(st):
                    np.testing.assert_allclose(s.numpy(), g[f"state{c}.{i}"], atol=2e-5, rtol=1e-4)
    for i, s in enumerate(st):
        np.testing.assert_allclose(s.numpy(), g[f"final_state.{i}"], atol=2e-5, rtol=1e-4)
    assert (st[-1].numpy() == 6 * chunk).all()
