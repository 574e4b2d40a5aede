"""RNN-T joiner (reference model/joiner/joiner.py:16-182).

`forward` keeps the reference's 4-tuple `(logits, boundary, ranges, simple_loss)`.  With
pruning and `use_out_project: false` (the zipformer YAML) the (B,T,R,C) lattice is NOT
materialised: `logits` is a `PrunedLattice` handle that Loss("Pruned_Rnnt") consumes with the
fused HIP joiner+loss kernel; `.materialize()` gives the tensor for any other consumer.
"""
import dataclasses
from typing import Tuple

import torch
import torch.nn as nn

from speech2text_amd import kernels as K
from speech2text_amd.model.layer.scaling import Linear


@dataclasses.dataclass
class JoinerConfig:
    input_dim: int
    output_dim: int
    inner_dim: int = 256
    activation: str = "relu"
    prune_range: int = 5
    lm_scale: float = 0.0
    am_scale: float = 0.0
    use_out_project: bool = True


class PrunedLattice:
    """Opaque stand-in for logits[b,t,i,:] = act(am[b,t,:] + lm[b,ranges[b,t,i],:])."""

    def __init__(self, am, lm, ranges, activation):
        self.am, self.lm, self.ranges, self.activation = am, lm, ranges, activation

    @property
    def shape(self):
        B, T, C = self.am.shape
        return torch.Size((B, T, self.ranges.shape[2], C))

    def size(self, i=None):
        return self.shape if i is None else self.shape[i]

    def materialize(self) -> torch.Tensor:
        B, T, C = self.am.shape
        R = self.ranges.shape[2]
        S1 = self.lm.shape[1]
        lm_p = torch.gather(self.lm.unsqueeze(1).expand(B, T, S1, C), 2,
                            self.ranges.reshape(B, T, R, 1).expand(B, T, R, C))
        x = self.am.unsqueeze(2) + lm_p
        return torch.relu(x) if self.activation == "relu" else torch.tanh(x)


class Joiner(nn.Module):
    def __init__(self, config: JoinerConfig) -> None:
        super().__init__()
        self._input_dim = config.input_dim
        self._output_dim = config.output_dim
        self._inner_dim = config.inner_dim
        self._enc_proj = Linear(self._input_dim, self._output_dim, bias=True)
        self._pre_proj = Linear(self._input_dim, self._output_dim, bias=True)
        if config.activation not in ("relu", "tanh"):
            raise ValueError(f"Unsupported activation {config.activation}")
        self._act_name = config.activation
        self._activation = nn.ReLU() if config.activation == "relu" else nn.Tanh()
        self._use_out_project = config.use_out_project
        if self._use_out_project:
            self._out_projection = nn.Sequential(Linear(self._output_dim, self._inner_dim),
                                                 Linear(self._inner_dim, self._output_dim))
        else:
            self._out_projection = nn.Identity()
        self._blank_token = 0
        self._prune_range = config.prune_range
        self._lm_scale = config.lm_scale
        self._am_scale = config.am_scale
        if self._lm_scale != 0.0 or self._am_scale != 0.0:
            raise NotImplementedError("lm_scale/am_scale != 0 (smoothed simple loss) is not on "
                                      "the accelerated path; every shipped YAML uses 0.0")

    @property
    def prune_range(self) -> int:
        return self._prune_range

    @property
    def blank_token(self) -> int:
        return self._blank_token

    def forward(self, encoder_out: torch.Tensor, encoder_out_lengths: torch.Tensor,
                predict_out: torch.Tensor, target_lengths: torch.Tensor,
                target: torch.Tensor = torch.empty(0, 0)):
        am = self._enc_proj(encoder_out)          # (B,T,C)
        lm = self._pre_proj(predict_out)          # (B,U+1,C)
        if self.prune_range > 0:
            assert target.shape[0] == target_lengths.shape[0] and target.dim() == 2
            dev = am.device
            boundary = K.make_boundary(target_lengths, encoder_out_lengths, dev)
            sym = target.to(device=dev, dtype=torch.int64).contiguous()
            neg_scores, px_grad, py_grad = K.rnnt_simple_loss(lm.float(), am.float(), sym,
                                                              boundary, self.blank_token)
            simple_loss = neg_scores.mean()
            ranges = K.rnnt_prune_ranges(px_grad, py_grad, boundary, self.prune_range)
            lattice = PrunedLattice(am, lm, ranges, self._act_name)
            if self._use_out_project:
                return self._out_projection(lattice.materialize()), boundary, ranges, simple_loss
            return lattice, boundary, ranges, simple_loss
        joint = am.unsqueeze(2).contiguous() + lm.unsqueeze(1).contiguous()
        return self._out_projection(self._activation(joint)), None, None, None

    def streaming_step(self, encoder_out: torch.Tensor, predictor_out: torch.Tensor):
        """(1,1,D) x (beam,1,D) -> (beam,V) log-probabilities (reference joiner.py:186-207)."""
        assert encoder_out.shape[0] == 1 and encoder_out.shape[1] == 1
        assert predictor_out.shape[1] == 1
        joint = self._enc_proj(encoder_out).unsqueeze(2) + self._pre_proj(predictor_out).unsqueeze(1)
        out = self._out_projection(self._activation(joint))
        return out.log_softmax(dim=-1).squeeze(1).squeeze(1)
