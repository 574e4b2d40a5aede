"""Oracle (test infrastructure only): numpy restatement of the reference's per-utterance
augmentation -- dataset/frontend/data_augmentation.py:13-56 AddNoise, :59-118 MixFeats,
:150-196 SpecAugment -- drawing from Python `random` in the reference's order.

PINNED by tests/golden/aug_ref.npz (outputs of the reference classes under fixed seeds,
tools/gen_golden.py gen_aug)."""
import random

import numpy as np


def spec_augment(feat, num_t_mask=2, num_f_mask=2, max_t=50, max_f=10):
    y = np.array(feat, copy=True)
    T, F = y.shape
    for _ in range(num_t_mask):
        start = random.randint(0, T - 1)
        length = random.randint(1, max_t)
        y[start:min(T, start + length), :] = 0
    for _ in range(num_f_mask):
        start = random.randint(0, F - 1)
        length = random.randint(1, max_f)
        y[:, start:min(F, start + length)] = 0
    return y


def _repeat_to(noise, n, axis):
    if n > noise.shape[axis]:
        reps = n // noise.shape[axis] + 1
        noise = np.concatenate([noise] * reps, axis=axis)
    return noise


def mix_feats(src, noise, snrs=(10, 20)):
    se, ne = float(np.exp(src).sum(dtype=np.float32)), float(np.exp(noise).sum(dtype=np.float32))
    snr = random.uniform(snrs[0], snrs[-1])
    gain = 1.0
    if se > 0.0 and ne > 0.0:
        gain = se * (10.0 ** (-snr / 10)) / ne
    noise = _repeat_to(noise, src.shape[0], 0)
    start = random.randint(0, noise.shape[0] - src.shape[0])
    nz = noise[start:start + src.shape[0]]
    return np.log(np.clip(np.exp(src) + np.float32(gain) * np.exp(nz), 1e-10, None)).astype(np.float32)


def add_noise(pcm, noise, min_snr_db=10, max_snr_db=50, max_gain_db=300.0):
    rms = lambda x: 10 * np.log10((x.astype(np.float32) ** 2).mean())   # noqa: E731
    snr = random.uniform(min_snr_db, max_snr_db)
    gain_db = min(rms(pcm) - rms(noise) - snr, max_gain_db)
    noise = noise * np.float32(10.0 ** (gain_db / 20.0))
    noise = _repeat_to(noise, pcm.shape[1], 1)
    start = random.randint(0, noise.shape[1] - pcm.shape[1])
    return np.clip(pcm + noise[:, start:start + pcm.shape[1]], -1.0, 1.0).astype(np.float32)
