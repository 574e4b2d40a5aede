"""Feature-pipeline factory (reference dataset/frontend/frontend.py:146-152).

Same `FeatType[feat_type].value(**feat_config)` surface, same per-utterance call
`forward(pcm[1,N]) -> feat[n,D]`; the arithmetic runs in the HIP fbank kernel.  Addition:
`forward_batch(pcm[B,Nmax], num_samples[B])` computes a whole padded batch in one launch
(optionally with the global CMVN fused), so features never leave HBM.
"""
from enum import Enum, unique

import torch
import torch.nn as nn

from speech2text_amd import kernels as K


class _HipFbank(nn.Module):
    def __init__(self, num_mel_bins, high_freq=0.0, low_freq=20.0, samplerate=16000,
                 scale_in=1.0):
        super().__init__()
        self._num_mel_bins = num_mel_bins
        self._high_freq, self._low_freq, self._samplerate = high_freq, low_freq, samplerate
        self._scale_in = scale_in
        self._tables = {}

    @property
    def pcm_normalize(self):
        return True

    @property
    def feat_dim(self):
        return self._num_mel_bins

    def tables(self, device):
        key = str(device)
        if key not in self._tables:
            self._tables[key] = K.FbankTables(self._num_mel_bins, float(self._samplerate),
                                              self._low_freq, self._high_freq, device=device)
        return self._tables[key]

    @torch.no_grad()
    def forward_batch(self, pcm, num_samples, cmvn_mean=None, cmvn_istd=None):
        dev = pcm.device if pcm.is_cuda else torch.device("cuda", torch.cuda.current_device())
        pcm = pcm.to(device=dev, dtype=torch.float32).contiguous()
        num_samples = num_samples.to(device=dev, dtype=torch.int64).contiguous()
        return K.fbank_batch(pcm, num_samples, self.tables(dev), cmvn_mean, cmvn_istd,
                             self._scale_in)

    @torch.no_grad()
    def forward(self, pcm: torch.Tensor) -> torch.Tensor:
        assert pcm.dim() == 2 and pcm.shape[0] == 1, "expects (1, num_samples)"
        n = torch.tensor([pcm.shape[1]], dtype=torch.int64)
        feats, _ = self.forward_batch(pcm, n)
        return feats[0]


class KaldiWaveFeature(_HipFbank):
    """`fbank`: torchaudio.compliance.kaldi.fbank semantics (reference :57-94)."""

    def __init__(self, num_mel_bins=64, frame_length=25, frame_shift=10, dither=0.0,
                 samplerate=16000) -> None:
        if frame_length != 25 or frame_shift != 10 or samplerate != 16000:
            raise NotImplementedError("the HIP fbank kernel is built for 25 ms / 10 ms @ 16 kHz "
                                      "(every shipped YAML)")
        if dither != 0.0:
            raise NotImplementedError("dither != 0 is not used by any shipped YAML")
        super().__init__(num_mel_bins, high_freq=0.0, samplerate=samplerate)
        self._frame_length, self._frame_shift, self._dither = frame_length, frame_shift, dither


class LhotseKaldiFeatFbank(_HipFbank):
    """`lhotes_fbank`: lhotse KaldifeatFbank (kaldifeat defaults: 80 bins whatever
    num_mel_bins says -- the reference only stores that argument, :104 --, high_freq=-400,
    dither 0, and 16-bit-scaled samples).  PARITY UNPINNED (lhotse/kaldifeat absent)."""

    def __init__(self, num_mel_bins=80, snip_edges=False) -> None:
        if not snip_edges:
            raise NotImplementedError("snip_edges=False framing is not on the accelerated path "
                                      "(the zipformer YAML sets snip_edges: true)")
        super().__init__(80, high_freq=-400.0, scale_in=32768.0)
        self._declared_mel_bins = num_mel_bins

    @property
    def feat_dim(self):
        return self._declared_mel_bins


class DummyFrontend(nn.Module):
    def __init__(self, dummy=-1) -> None:
        super().__init__()

    @property
    def pcm_normalize(self):
        return True

    @property
    def feat_dim(self):
        return -1

    @torch.no_grad()
    def forward(self, pcm: torch.Tensor) -> torch.Tensor:
        return pcm.squeeze(0)


class TorchScriptKaldiWaveFeature(nn.Module):
    def __init__(self, torchscript: str, num_mel_bins=80) -> None:
        super().__init__()
        self._frontend_sess = torch.jit.load(torchscript)
        self._num_mel_bins = num_mel_bins

    @property
    def pcm_normalize(self):
        return True

    @property
    def feat_dim(self):
        return self._num_mel_bins

    @torch.no_grad()
    def forward(self, pcm: torch.Tensor) -> torch.Tensor:
        return self._frontend_sess(pcm)


@unique
class FeatType(Enum):
    pcm = DummyFrontend
    fbank = KaldiWaveFeature
    lhotes_fbank = LhotseKaldiFeatFbank
    torchscript_fbank = TorchScriptKaldiWaveFeature
