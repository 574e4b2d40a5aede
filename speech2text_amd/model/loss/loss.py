"""Loss factory + wrappers (reference model/loss/{loss,ctc_loss,rnnt_loss,pruned_rnnt_loss,
cross_entropy,kl_divergence}.py).  CTC / RNN-T / pruned RNN-T run the HIP lattice kernels."""
import dataclasses
import math
from typing import Dict

import torch
import torch.nn as nn

from speech2text_amd import kernels as K
from speech2text_amd.model.functions.masking import make_non_pad_mask
from speech2text_amd.model.joiner.joiner import PrunedLattice


@dataclasses.dataclass
class CtcLossConfig:
    blank_label: int = 0
    reduction: str = "mean"
    zero_infinity: bool = True


class CtcLoss(nn.Module):
    def __init__(self, config: CtcLossConfig):
        super().__init__()
        self._blank_label = config.blank_label
        self._reduction = config.reduction
        self._zero_infinity = config.zero_infinity

    def forward(self, logits, targets, logits_length, targets_length):
        return K.ctc_loss(logits, targets, logits_length, targets_length, self._blank_label,
                          self._reduction, self._zero_infinity)


@dataclasses.dataclass
class RnntLossConfig:
    blank_label: int = 0
    clamp: float = -1
    reduction: str = "mean"


def _reduce(per_utt, reduction):
    if reduction == "mean":
        return per_utt.mean()
    if reduction == "sum":
        return per_utt.sum()
    return per_utt


class RnntLoss(nn.Module):
    """Full-lattice transducer loss (torchaudio RNNTLoss semantics: log_softmax inside)."""

    def __init__(self, config: RnntLossConfig) -> None:
        super().__init__()
        if config.clamp is not None and config.clamp > 0:
            raise NotImplementedError("gradient clamp > 0 is not used by any shipped YAML")
        self._blank = config.blank_label
        self._reduction = config.reduction

    def forward(self, logits, targets, logits_length, targets_length):
        dev = logits.device
        boundary = K.make_boundary(targets_length, logits_length, dev)
        sym = targets.to(device=dev, dtype=torch.int64).contiguous()
        return _reduce(K.rnnt_lattice_loss(logits, None, sym, boundary, self._blank),
                       self._reduction)


@dataclasses.dataclass
class PrunedRnntLossConfig:
    termination_symbol: int = 0
    rnnt_type: str = "regular"
    delay_penalty: float = 0.0
    reduction: str = "mean"


class PrunedRnntLoss(nn.Module):
    def __init__(self, config: PrunedRnntLossConfig) -> None:
        super().__init__()
        if config.rnnt_type != "regular" or config.delay_penalty != 0.0:
            raise NotImplementedError("only rnnt_type='regular', delay_penalty=0 (all shipped "
                                      "YAMLs) are on the accelerated path")
        self._termination_symbol = config.termination_symbol
        self._reduction = config.reduction

    def forward(self, logits, targets, logits_length, targets_length, boundary, ranges):
        sym = targets.to(device=ranges.device, dtype=torch.int64).contiguous()
        if isinstance(logits, PrunedLattice):
            per = K.rnnt_pruned_joiner_loss(logits.am, logits.lm, ranges, sym, boundary,
                                            self._termination_symbol, logits.activation)
        else:
            per = K.rnnt_lattice_loss(logits, ranges, sym, boundary, self._termination_symbol)
        return _reduce(per, self._reduction)


@dataclasses.dataclass
class MaskedCELossConfig:
    num_classes: int = 1025
    scale_factor: float = 1.0
    label_smoothing: float = 0.0


class MaskedCELoss(nn.Module):
    """Cross entropy with label smoothing and a frame mask (reference cross_entropy.py:38-69),
    computed by the fused row kernel: target = eps/K everywhere + (1-eps) at the label."""

    def __init__(self, config: MaskedCELossConfig):
        super().__init__()
        self._num_classes = config.num_classes
        self._scale_factor = config.scale_factor
        self._label_smoothing = config.label_smoothing

    def forward(self, logits, ori_labels, mask=None):
        max_len = logits.size(1)
        nc = self._num_classes
        eps = float(self._label_smoothing)
        row = K.smoothed_nll_rows(logits.contiguous().reshape(-1, nc), ori_labels.reshape(-1),
                                  self._scale_factor, eps / nc, 1.0 - eps + eps / nc, 0.0)
        if mask is not None:
            if mask.dim() == 1:
                assert int(mask.max()) == max_len
                mask = make_non_pad_mask(mask)
            mask = mask.contiguous().reshape(-1).to(row.dtype)
            return (row * mask).sum() / mask.sum()
        return row.mean()

    def predict(self, logits):
        logits = logits * self._scale_factor
        return logits.softmax(dim=-1)


@dataclasses.dataclass
class MaskedKLDivergenceConfig:
    num_classes: int = 1025
    scale_factor: float = 1.0
    label_smoothing: float = 0.0


class MaskedKLDivergence(nn.Module):
    def __init__(self, config: MaskedKLDivergenceConfig) -> None:
        super().__init__()
        self._num_classes = config.num_classes
        self._scale_factor = config.scale_factor
        self._label_smoothing = config.label_smoothing

    def forward(self, logits, ori_labels, mask=None):
        """KL(smoothed one-hot || softmax) summed over classes, masked mean over frames
        (reference kl_divergence.py:36-76).  The smoothed-label tensor, the log-softmax and the
        element-wise KL are never materialised (fused row kernel).  The reference scales
        `logits` IN PLACE by scale_factor on every call (:61), so its second call per codebook
        sees scale_factor**2; that quirk is not reproduced (every shipped YAML uses 1.0)."""
        nc = self._num_classes
        if mask is not None:
            if mask.dim() == 1:
                assert int(mask.max()) == logits.size(1)
                mask = make_non_pad_mask(mask)
            mask = mask.contiguous().reshape(-1)
        else:
            mask = torch.ones_like(ori_labels).reshape(-1)
        a = self._label_smoothing / (nc - 1)
        b = 1.0 - self._label_smoothing
        c0 = (nc - 1) * (a * math.log(a) if a > 0 else 0.0) + (b * math.log(b) if b > 0 else 0.0)
        row = K.smoothed_nll_rows(logits.contiguous().reshape(-1, nc), ori_labels.reshape(-1),
                                  self._scale_factor, a, b, c0)
        m = mask.to(row.dtype)
        return (row * m).sum() / m.sum()

    def predict(self, logits):
        logits = logits * self._scale_factor
        return logits.log_softmax(dim=-1)


class Loss(nn.Module):
    def __init__(self, config) -> None:
        super().__init__()
        name = config["model"]
        if name == "CTC":
            self.loss = CtcLoss(CtcLossConfig(**config["config"]))
        elif name == "Rnnt":
            self.loss = RnntLoss(RnntLossConfig(**config["config"]))
        elif name == "Pruned_Rnnt":
            self.loss = PrunedRnntLoss(PrunedRnntLossConfig(**config["config"]))
        elif name == "MaskedCELoss":
            self.loss = MaskedCELoss(MaskedCELossConfig(**config["config"]))
        elif name == "MaskedKLDiv":
            self.loss = MaskedKLDivergence(MaskedKLDivergenceConfig(**config["config"]))
        else:
            raise ValueError("Not support {} loss".format(name))

    def forward(self, batch: Dict[str, torch.Tensor]):
        return self.loss(**batch)

    def predict(self, logits: torch.Tensor):
        if hasattr(self.loss, "predict"):
            return self.loss.predict(logits)
