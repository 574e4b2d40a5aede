"""Zipformer support layers (host-side mirror of the reference's model/layer/scaling.py).

Same class names, constructor arguments and state_dict keys as the reference so YAML
configs and checkpoints interchange; the implementations are our own: forward passes and
the gradient-shaping backward passes (Balancer, Whiten) are closed-form GPU code instead of
autograd-inside-backward (reference: scaling.py:741-789, 994-1028).
"""
import math
import random
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from speech2text_amd import rng
from speech2text_amd import zip_kernels as zk


# ------------------------------------------------------------------ ScheduledFloat
class ScheduledFloat(nn.Module):
    """Piecewise-linear schedule over `batch_count`; float(x) gives `default` while
    batch_count is None or the module is in eval mode (reference scaling.py:161-217; the
    reference's training loop never sets batch_count, so defaults rule)."""

    def __init__(self, *pairs, default: float = 0.0):
        super().__init__()
        self.pairs = [(float(x), float(y)) for x, y in pairs]
        self.batch_count = None
        self.name = None
        self.default = default

    def value_at(self, x: float) -> float:
        p = self.pairs
        if x <= p[0][0]:
            return p[0][1]
        if x >= p[-1][0]:
            return p[-1][1]
        for (x0, y0), (x1, y1) in zip(p[:-1], p[1:]):
            if x0 <= x <= x1:
                return y0 + (y1 - y0) * (x - x0) / (x1 - x0)
        return p[-1][1]

    def __float__(self):
        if self.batch_count is None or not self.training:
            return float(self.default)
        return self.value_at(self.batch_count)

    def extra_repr(self):
        return f"batch_count={self.batch_count}, pairs={self.pairs}, default={self.default}"


FloatLike = Union[float, ScheduledFloat]


_REPLAY = []


def _rand() -> float:
    """random.random(), the generator every gradient-shaping decision of the reference draws from
    (scaling.py:836,1071,1186).  The layer executor (speech2text_amd/zip_layer.py) draws a layer's
    decisions up front; when it hands the layer back to the module-by-module path it queues the
    values it already drew here so that path consumes the very same stream."""
    return _REPLAY.pop(0) if _REPLAY else random.random()


def _no_op(x: Tensor) -> Tensor:
    """Identity.  (The reference returns `x.chunk(1, dim=-1)[0]`, scaling.py:1193-1199, to hand
    TorchScript a distinct tensor; that costs an autograd node per non-firing Balancer / Whiten
    -- ~200 per step -- and changes no value, so the tensor itself is returned.)"""
    return x


class Identity(nn.Module):
    def forward(self, x):
        return _no_op(x)


# ------------------------------------------------------------------ gradient shaping
class _BalancerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cfg):
        ctx.save_for_backward(x)
        ctx.cfg = cfg
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return zk.balancer_backward(x, g, *ctx.cfg), None


class _BalancerSwooshFn(torch.autograd.Function):
    """swoosh(balancer(x)): the Balancer is the identity in forward, so this is the activation;
    backward takes the gradient through the activation and the Balancer's update in ONE pass
    (s2t_balancer_bwd with act_off) when the Balancer fired, else through the activation only."""

    @staticmethod
    def forward(ctx, x, cfg, is_l):
        ctx.save_for_backward(x)
        ctx.cfg, ctx.is_l = cfg, is_l
        return zk.swoosh_forward(x, is_l)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        if ctx.cfg is None:
            return zk.swoosh_backward(x, g, ctx.is_l), None, None
        return zk.balancer_backward(x, g.contiguous(), *ctx.cfg, swoosh_l=ctx.is_l), None, None


def balancer_swoosh(balancer: "Balancer", x: Tensor, is_l: bool) -> Tensor:
    """SwooshL/R(balancer(x)) with the Balancer's random draw made here (as its forward would)."""
    fires = balancer.fires(x)
    return _BalancerSwooshFn.apply(x, balancer.cfg(x.ndim) if fires else None, is_l)


def _prop_pos_to_mean(x: float) -> float:
    x = -1 + 2 * x
    eps = 1.0e-10
    return 0.8139535143 * (math.log(1 + x + eps) - math.log(1 - x + eps)) / 2.0


class Balancer(nn.Module):
    """Identity in forward; in backward adds a per-channel gradient term that pushes the
    channel's mean/rms back inside [min_positive,max_positive] / [min_abs,max_abs]
    (reference scaling.py:792-902).  The reference's GPU-memory cutoff heuristic
    (scaling.py:842,854-856) does not exist on its CPU path and is not reproduced."""

    def __init__(self, num_channels: int, channel_dim: int, min_positive: FloatLike = 0.05,
                 max_positive: FloatLike = 0.95, min_abs: FloatLike = 0.2,
                 max_abs: FloatLike = 100.0, grad_scale: FloatLike = 0.04,
                 prob: Optional[FloatLike] = None):
        super().__init__()
        if prob is None:
            prob = ScheduledFloat((0.0, 0.5), (8000.0, 0.125), default=0.4)
        self.prob = prob
        self.num_channels = num_channels
        self.channel_dim = channel_dim
        self.min_positive = min_positive
        self.max_positive = max_positive
        self.min_abs = min_abs
        self.max_abs = max_abs
        self.grad_scale = grad_scale

    def fires(self, x: Tensor) -> bool:
        """Draws this call's random decision (one random.random(), as the reference)."""
        return x.requires_grad and _rand() < float(self.prob)

    def cfg(self, ndim: int):
        """(min_mean, max_mean, min_rms, max_rms, grad_scale, channel_dim) for the backward."""
        return (_prop_pos_to_mean(float(self.min_positive)),
                _prop_pos_to_mean(float(self.max_positive)),
                1.25331413732 * float(self.min_abs), 1.25331413732 * float(self.max_abs),
                float(self.grad_scale), self.channel_dim % ndim)

    def shape_grad(self, x: Tensor) -> Tensor:
        assert x.shape[self.channel_dim] == self.num_channels
        return _BalancerFn.apply(x, self.cfg(x.ndim))

    def forward(self, x: Tensor) -> Tensor:
        return self.shape_grad(x) if self.fires(x) else _no_op(x)


class _WhitenFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, module):
        ctx.save_for_backward(x)
        ctx.module = module
        ctx.stats = zk.WhitenStats(x, module.num_groups)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        w = ctx.module
        out, active = zk.whiten_backward(x, g, ctx.stats, float(w.whitening_limit),
                                         float(w.grad_scale))
        w.prob = w.max_prob if active else w.min_prob
        return out, None


class Whiten(nn.Module):
    """Identity in forward; backward adds the gradient of the whitening metric when it
    exceeds `whitening_limit` (reference scaling.py:949-1095)."""

    def __init__(self, num_groups: int, whitening_limit: FloatLike,
                 prob: Union[float, Tuple[float, float]], grad_scale: FloatLike):
        super().__init__()
        assert num_groups >= 1 and float(whitening_limit) >= 1 and float(grad_scale) >= 0
        self.num_groups = num_groups
        self.whitening_limit = whitening_limit
        self.grad_scale = grad_scale
        if isinstance(prob, float):
            prob = (prob, prob)
        self.min_prob, self.max_prob = prob
        assert 0 < self.min_prob <= self.max_prob <= 1
        self.prob = self.max_prob
        self.name = None

    def fires(self, x: Tensor) -> bool:
        return not (not x.requires_grad or _rand() > self.prob
                    or float(self.grad_scale) == 0)

    def shape_grad(self, x: Tensor) -> Tensor:
        return _WhitenFn.apply(x, self)

    def forward(self, x: Tensor) -> Tensor:
        return self.shape_grad(x) if self.fires(x) else _no_op(x)


class _LimitParamFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lo, hi):
        ctx.save_for_backward(x)
        ctx.lo, ctx.hi = lo, hi
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return zk.limit_param_grad(x, g, ctx.lo, ctx.hi), None, None


def limit_param_value(x: Tensor, min: float, max: float, prob: float = 0.6,
                      training: bool = True) -> Tensor:
    """Flips the gradient sign of parameters that are outside [min,max] and moving further
    out (reference scaling.py:1153-1190)."""
    if training and _rand() < prob:
        return _LimitParamFn.apply(x, min, max)
    return x


class _AbsPenaltyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, limit, penalty):
        ctx.save_for_backward(x)
        ctx.limit, ctx.penalty = limit, penalty
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        # d/dx of penalty * sum(|x| over elements with |x| > limit)
        return g + ctx.penalty * torch.where(x.abs() > ctx.limit, x.sign(), torch.zeros_like(x)), \
            None, None


def penalize_abs_values_gt(x: Tensor, limit: float, penalty: float, name: str = None) -> Tensor:
    """Reference scaling.py:905-935 (with_loss of penalty*sign(x)*[|x|>limit]*x)."""
    return _AbsPenaltyFn.apply(x, limit, penalty)


class _ScaleGradFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, alpha):
        ctx.alpha = alpha
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.alpha, None


class ScaleGrad(nn.Module):
    def __init__(self, alpha: float):
        super().__init__()
        self.alpha = alpha

    def forward(self, x: Tensor) -> Tensor:
        if not self.training:
            return x
        return _ScaleGradFn.apply(x, self.alpha)


def softmax(x: Tensor, dim: int) -> Tensor:
    return x.softmax(dim=dim)


# ------------------------------------------------------------------ norms / linear helpers
class BiasNorm(nn.Module):
    """x * (mean((x-bias)^2))^-0.5 * exp(log_scale)   (reference scaling.py:347-476)."""

    def __init__(self, num_channels: int, channel_dim: int = -1, log_scale: float = 1.0,
                 log_scale_min: float = -1.5, log_scale_max: float = 1.5,
                 store_output_for_backprop: bool = False):
        super().__init__()
        self.num_channels = num_channels
        self.channel_dim = channel_dim
        self.log_scale = nn.Parameter(torch.tensor(log_scale))
        self.bias = nn.Parameter(torch.empty(num_channels).normal_(mean=0, std=1e-4))
        self.log_scale_min = log_scale_min
        self.log_scale_max = log_scale_max

    def forward(self, x: Tensor) -> Tensor:
        assert x.shape[self.channel_dim] == self.num_channels
        log_scale = limit_param_value(self.log_scale, min=float(self.log_scale_min),
                                      max=float(self.log_scale_max), training=self.training)
        if self.channel_dim in (-1, x.ndim - 1):
            return zk.bias_norm(x, self.bias, log_scale)
        xt = x.transpose(self.channel_dim, -1)
        return zk.bias_norm(xt, self.bias, log_scale).transpose(self.channel_dim, -1)


class Linear(nn.Linear):
    """nn.Linear (same parameters, same init) whose weight/bias gradients come from the
    split-row MFMA kernel (zip_kernels.linear_wgrad) when the shape is tall."""

    def forward(self, x: Tensor) -> Tensor:
        return zk.linear(x, self.weight, self.bias)


def ScaledLinear(*args, initial_scale: float = 1.0, **kwargs) -> nn.Linear:
    ans = Linear(*args, **kwargs)
    with torch.no_grad():
        ans.weight[:] *= initial_scale
        if ans.bias is not None:
            torch.nn.init.uniform_(ans.bias, -0.1 * initial_scale, 0.1 * initial_scale)
    return ans


def ScaledConv2d(*args, initial_scale: float = 1.0, **kwargs) -> nn.Conv2d:
    ans = nn.Conv2d(*args, **kwargs)
    with torch.no_grad():
        ans.weight[:] *= initial_scale
        if ans.bias is not None:
            torch.nn.init.uniform_(ans.bias, -0.1 * initial_scale, 0.1 * initial_scale)
    return ans


class ChunkCausalDepthwiseConv1d(nn.Module):
    """Causal half-kernel depthwise conv + within-chunk full-kernel depthwise conv scaled by a
    per-position edge factor (reference scaling.py:552-681).  Parameters keep the reference's
    names/shapes; the computation is one fused HIP kernel working directly on (T,B,C)."""

    def __init__(self, channels: int, kernel_size: int, initial_scale: float = 1.0,
                 bias: bool = True):
        super().__init__()
        assert kernel_size % 2 == 1
        half = (kernel_size + 1) // 2
        self.causal_conv = nn.Conv1d(channels, channels, groups=channels, kernel_size=half,
                                     padding=0, bias=True)
        self.chunkwise_conv = nn.Conv1d(channels, channels, groups=channels,
                                        kernel_size=kernel_size, padding=kernel_size // 2,
                                        bias=bias)
        self.chunkwise_conv_scale = nn.Parameter(torch.zeros(2, channels, kernel_size))
        self.kernel_size = kernel_size
        with torch.no_grad():
            self.causal_conv.weight[:] *= initial_scale
            self.chunkwise_conv.weight[:] *= initial_scale
            if bias:
                torch.nn.init.uniform_(self.causal_conv.bias, -0.1 * initial_scale,
                                       0.1 * initial_scale)

    def forward(self, x: Tensor, chunk_size: int = -1) -> Tensor:
        """x: (batch, channels, time) as in the reference."""
        y = zk.glu_chunk_causal_dwconv(x.permute(2, 0, 1).contiguous(), None, None, self,
                                       chunk_size)
        return y.permute(1, 2, 0)


# ------------------------------------------------------------------ activations
class SwooshL(nn.Module):
    def forward(self, x: Tensor) -> Tensor:
        return zk.swoosh(x, True)


class SwooshR(nn.Module):
    def forward(self, x: Tensor) -> Tensor:
        return zk.swoosh(x, False)


class Dropout2(nn.Module):
    def __init__(self, p: FloatLike):
        super().__init__()
        self.p = p

    def forward(self, x: Tensor) -> Tensor:
        return F.dropout(x, p=float(self.p), training=self.training)


class Dropout3(nn.Module):
    """Dropout whose mask is shared across `shared_dim` (reference scaling.py:1319-1337)."""

    def __init__(self, p: FloatLike, shared_dim: int):
        super().__init__()
        self.p = p
        self.shared_dim = shared_dim

    def forward(self, x: Tensor) -> Tensor:
        p = float(self.p)
        if not self.training or p == 0:
            return _no_op(x)
        shape = list(x.shape)
        shape[self.shared_dim] = 1
        mask = (rng.rand(*shape, device=x.device) > p).to(x.dtype) * (1.0 / (1 - p))
        return x * mask


class ActivationDropoutAndLinear(nn.Module):
    """Swoosh activation -> (shared-mask) dropout -> Linear, storing only its input for the
    backward (reference scaling.py:1512-1668)."""

    def __init__(self, in_channels: int, out_channels: int, bias: bool = True,
                 activation: str = "SwooshL", dropout_p: FloatLike = 0.0,
                 dropout_shared_dim: Optional[int] = -1, initial_scale: float = 1.0):
        super().__init__()
        l = ScaledLinear(in_channels, out_channels, bias=bias, initial_scale=initial_scale)
        self.weight = l.weight
        self.register_parameter("bias", l.bias)
        assert activation in ("SwooshL", "SwooshR")
        self.activation = activation
        self.dropout_p = dropout_p
        self.dropout_shared_dim = dropout_shared_dim

    def forward(self, x: Tensor, residual: Optional[Tensor] = None) -> Tensor:
        p = float(self.dropout_p) if self.training else 0.0
        mask = None
        if p != 0.0:
            shape = list(x.shape)
            if self.dropout_shared_dim is not None:
                shape[self.dropout_shared_dim] = 1
            mask = (1.0 / (1.0 - p)) * (rng.rand(*shape, device=x.device, dtype=x.dtype) > p)
        return zk.swoosh_linear(x, self.weight, self.bias, self.activation == "SwooshL", mask,
                                residual)


def convert_num_channels(x: Tensor, num_channels: int) -> Tensor:
    if num_channels <= x.shape[-1]:
        return x[..., :num_channels]
    pad = list(x.shape)
    pad[-1] = num_channels - x.shape[-1]
    return torch.cat((x, torch.zeros(pad, dtype=x.dtype, device=x.device)), dim=-1)
