"""CPU: the oracle restatements against golden vectors produced by the reference."""
import os

import numpy as np
import pytest

from oracle import ctc as octc
from oracle import fbank as ofbank


def test_fbank_oracle_vs_torchscript_archive(golden_dir):
    g = np.load(os.path.join(golden_dir, "fbank_script64.npz"))
    for i in range(6):
        got = ofbank.fbank(g[f"pcm{i}"], num_mel_bins=64)
        ref = g[f"feat{i}"]
        assert got.shape == ref.shape
        # fp32 FFT implementations differ in the last bits; log() amplifies near-eps bins
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-3)
        assert np.abs(got - ref).mean() < 2e-5


def test_fbank_framing_law():
    # dataset/frontend/frontend_test.py:83-103: 12560 samples -> 77 frames, hop 160
    rng = np.random.default_rng(0)
    x = rng.standard_normal(12560 * 2 - 240).astype(np.float32) * 0.1
    a = ofbank.fbank(x[:12560], 80)
    assert a.shape == (77, 80)
    full = ofbank.fbank(x, 80)
    b = ofbank.fbank(x[12560 - 240:], 80)
    np.testing.assert_allclose(full[77:77 + b.shape[0]], b, atol=1e-4)


@pytest.mark.parametrize("ci", [0, 1, 2, 3])
def test_ctc_oracle_vs_reference_module(golden_dir, ci):
    g = np.load(os.path.join(golden_dir, "ctc_ref.npz"))
    loss, grad, _ = octc.ctc_loss(g[f"logits{ci}"], g[f"targets{ci}"], g[f"in_len{ci}"],
                                  g[f"tgt_len{ci}"])
    np.testing.assert_allclose(loss, g[f"loss{ci}"], rtol=2e-6)
    # the reference runs the recursion in fp32; the oracle in fp64
    np.testing.assert_allclose(grad, g[f"grad{ci}"], atol=1e-5, rtol=2e-4)
