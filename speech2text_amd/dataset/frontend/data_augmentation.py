"""Data augmentation on the GPU, batched (mirror of the reference's
dataset/frontend/data_augmentation.py: AddNoise :13-56, MixFeats :59-118, SpecAugment :150-196;
SpeedPerturb :121-147 is sox and stays out of scope).

Same class names, constructor arguments and per-utterance random draws (Python `random`, in the
reference's order) -- but the reference processes one utterance per call on a CPU DataLoader
worker, while `process_batch` here applies the op to the whole padded device batch in one HIP
launch (csrc/augment.hip) right after / before the on-GPU fbank."""
import random

import torch

from speech2text_amd import _native as N


def _dev(*ts):
    for t in ts:
        if not t.is_cuda:
            raise RuntimeError("speech2text_amd augmentation runs on the GPU only (no CPU fallback)")


def _i64(v, dev):
    return torch.as_tensor(v, dtype=torch.int64, device=dev).contiguous()


def _energy(x, lens, D, mode):
    out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    N.check(N.lib().s2t_row_energy(N.fp(x), x.stride(0), N.lp(lens), x.shape[0], D, mode, N.fp(out),
                                   N.stream()), "s2t_row_energy")
    return out


def _mix(src, slen, noise, nlen, start, snr, D, mode, max_gain_db=0.0):
    src = src.contiguous().float()
    noise = noise.contiguous().float()
    dev = src.device
    slen, nlen, start = _i64(slen, dev), _i64(nlen, dev), _i64(start, dev)
    snr = torch.as_tensor(snr, dtype=torch.float32, device=dev).contiguous()
    se = _energy(src, slen, D, mode)
    ne = _energy(noise, nlen, D, mode)
    out = torch.empty_like(src)
    rows_max = src.shape[1]
    N.profile_note("s2t_mix", 4.0 * (2 * src.numel() + noise.numel()))
    N.check(N.lib().s2t_mix(N.fp(src), src.stride(0), N.lp(slen), N.fp(noise), noise.stride(0),
                            N.lp(nlen), N.lp(start), N.fp(snr), N.fp(se), N.fp(ne), src.shape[0],
                            rows_max, D, mode, float(max_gain_db), N.fp(out), N.stream()), "s2t_mix")
    return out


class AddNoise(object):
    """pcm + noise scaled to a random SNR, clipped to [-1, 1] (reference :13-56)."""

    def __init__(self, min_snr_db=10, max_snr_db=50, max_gain_db=300.0) -> None:
        self._min_snr_db, self._max_snr_db, self._max_gain_db = min_snr_db, max_snr_db, max_gain_db

    def draw(self, pcm_len: int, noise_len: int):
        """The reference's two draws for one utterance: (snr_db, start)."""
        snr = random.uniform(self._min_snr_db, self._max_snr_db)
        total = noise_len * (pcm_len // noise_len + 1) if pcm_len > noise_len else noise_len
        return snr, random.randint(0, total - pcm_len)

    def process_batch(self, pcm, pcm_len, noise, noise_len, draws=None):
        """pcm (B,Nmax), noise (B,Mmax) device tensors, lengths (B) -> augmented (B,Nmax)."""
        _dev(pcm, noise)
        pl, nl = [int(v) for v in pcm_len], [int(v) for v in noise_len]
        if draws is None:
            draws = [self.draw(a, b) for a, b in zip(pl, nl)]
        return _mix(pcm, pl, noise, nl, [d[1] for d in draws], [d[0] for d in draws], 1, 1,
                    self._max_gain_db)

    def process(self, pcm: torch.Tensor, noise_pcm: torch.Tensor) -> torch.Tensor:
        """(1,N), (1,M) -> (1,N), as the reference's per-utterance call."""
        return self.process_batch(pcm, [pcm.shape[1]], noise_pcm, [noise_pcm.shape[1]])


class MixFeats(object):
    """log(exp(src) + gain * exp(noise)) at a random SNR (reference :59-118)."""

    def __init__(self, snrs=(10, 20)) -> None:
        self._snrs = snrs

    def draw(self, src_len: int, noise_len: int):
        snr = random.uniform(self._snrs[0], self._snrs[-1])
        total = noise_len * (src_len // noise_len + 1) if src_len > noise_len else noise_len
        return snr, random.randint(0, total - src_len)

    def process_batch(self, src, src_len, noise, noise_len, draws=None):
        """src (B,T,D), noise (B,Tn,D) device log-mel features -> mixed (B,T,D)."""
        _dev(src, noise)
        sl, nl = [int(v) for v in src_len], [int(v) for v in noise_len]
        if draws is None:
            draws = [self.draw(a, b) for a, b in zip(sl, nl)]
        return _mix(src, sl, noise, nl, [d[1] for d in draws], [d[0] for d in draws],
                    src.shape[-1], 0)

    def process(self, src: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
        return self.process_batch(src.unsqueeze(0), [src.shape[0]], noise.unsqueeze(0),
                                  [noise.shape[0]])[0]


class SpecAugment(object):
    """num_t_mask time spans and num_f_mask frequency spans set to zero (reference :150-196)."""

    def __init__(self, num_t_mask=2, num_f_mask=2, max_t=50, max_f=10, max_w=80) -> None:
        self._num_t_mask, self._num_f_mask = num_t_mask, num_f_mask
        self._max_t, self._max_f, self._max_w = max_t, max_f, max_w

    def draw(self, frames: int, freq: int):
        ts, fs = [], []
        for _ in range(self._num_t_mask):
            start = random.randint(0, frames - 1)
            length = random.randint(1, self._max_t)
            ts.append((start, min(frames, start + length)))
        for _ in range(self._num_f_mask):
            start = random.randint(0, freq - 1)
            length = random.randint(1, self._max_f)
            fs.append((start, min(freq, start + length)))
        return ts, fs

    def process_batch(self, feats, feat_len, draws=None):
        """feats (B,T,F) device tensor -> augmented copy (the reference clones too)."""
        _dev(feats)
        y = feats.detach().clone().contiguous().float()
        B, T, F = y.shape
        if draws is None:
            draws = [self.draw(int(n), F) for n in feat_len]
        nt, nf = self._num_t_mask, self._num_f_mask
        tspan = torch.tensor([d[0] for d in draws], dtype=torch.int32).reshape(B, nt, 2).to(y.device)
        fspan = torch.tensor([d[1] for d in draws], dtype=torch.int32).reshape(B, nf, 2).to(y.device)
        N.profile_note("s2t_specaug", 8.0 * y.numel())
        N.check(N.lib().s2t_specaug(N.fp(y), B, T, F, N.ip(tspan.contiguous()), nt,
                                    N.ip(fspan.contiguous()), nf, N.stream()), "s2t_specaug")
        return y

    def process(self, feat: torch.Tensor) -> torch.Tensor:
        return self.process_batch(feat.unsqueeze(0), [feat.shape[0]])[0]
