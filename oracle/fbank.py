"""Oracle: kaldi-style fbank, numpy float32 restatement.

Follows dataset/frontend/frontend.py:85-94 -> torchaudio.compliance.kaldi.fbank
(torchaudio 0.13.1, not vendored) as stated op-by-op by the TorchScript archive
sample_data/model/frontend.script (the reference's own exported frontend).
PINNED: tests/golden/fbank_*.npz hold that archive's outputs (64 mel bins) on
seeded PCM; tools/gen_golden.py made them.  Other bin counts / high_freq use the
same code path with different constants (parity for those: unpinned).
"""
import math

import numpy as np

EPS = np.float32(1.1920928955078125e-07)


def mel_scale(f):
    return 1127.0 * math.log(1.0 + f / 700.0)


def povey_window():
    # hann_window(400, periodic=False) ** 0.85 in float32
    n = np.arange(400, dtype=np.float64)
    hann = (0.5 - 0.5 * np.cos(2.0 * math.pi * n / 399.0)).astype(np.float32)
    return np.power(hann, np.float32(0.85)).astype(np.float32)


def mel_banks(num_bins, sample_freq=16000.0, low_freq=20.0, high_freq=0.0, nfft=512):
    """(num_bins, nfft/2+1) float32, last column zero (frontend.script: mel_energies1)."""
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    fft_bin_width = sample_freq / nfft
    mel_low = mel_scale(low_freq)
    mel_high = mel_scale(high_freq)
    delta = (mel_high - mel_low) / (num_bins + 1)
    b = np.arange(num_bins, dtype=np.float32)[:, None]
    d32, l32 = np.float32(delta), np.float32(mel_low)
    left = b * d32 + l32
    center = (b + np.float32(1.0)) * d32 + l32
    right = (b + np.float32(2.0)) * d32 + l32
    freq = np.arange(nfft // 2, dtype=np.float32) * np.float32(fft_bin_width)
    mel = (np.log(freq / np.float32(700.0) + np.float32(1.0)) * np.float32(1127.0))[None, :]
    mel = mel.astype(np.float32)
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    w = np.maximum(np.float32(0.0), np.minimum(up, down)).astype(np.float32)
    return np.pad(w, ((0, 0), (0, 1))).astype(np.float32)


def num_frames(num_samples):
    return 0 if num_samples < 400 else 1 + (num_samples - 400) // 160


def fbank(pcm, num_mel_bins=80, high_freq=0.0, low_freq=20.0):
    """pcm: (N,) float32 in [-1,1] (pcm_normalize=True) -> (n, num_mel_bins) float32."""
    pcm = np.asarray(pcm, dtype=np.float32).reshape(-1)
    m = num_frames(pcm.shape[0])
    if m == 0:
        return np.zeros((0, num_mel_bins), np.float32)
    idx = np.arange(m)[:, None] * 160 + np.arange(400)[None, :]
    fr = pcm[idx]                                            # as_strided [m,400],[160,1]
    fr = fr - fr.mean(axis=1, keepdims=True, dtype=np.float32)
    prev = np.concatenate([fr[:, :1], fr[:, :-1]], axis=1)   # replicate pad
    fr = fr - prev * np.float32(0.97)
    fr = fr * povey_window()[None, :]
    fr = np.pad(fr, ((0, 0), (0, 112)))
    spec = np.fft.rfft(fr.astype(np.float32), axis=1)
    power = (np.abs(spec).astype(np.float32)) ** np.float32(2.0)
    mel = power.astype(np.float32) @ mel_banks(num_mel_bins, high_freq=high_freq,
                                               low_freq=low_freq).T
    return np.log(np.maximum(mel.astype(np.float32), EPS)).astype(np.float32)
