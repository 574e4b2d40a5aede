"""Zipformer frontend: Conv2dSubsampling + ConvNeXt (mirror of the reference's
model/layer/subsampling.py; same parameter names, T' = (T-7)//2)."""
from typing import Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from speech2text_amd import rng
from speech2text_amd import zip_kernels as zk

from speech2text_amd.model.layer.scaling import (Linear, Balancer, BiasNorm, Dropout3, FloatLike,
                                                 ScaledConv2d, ScaleGrad, ScheduledFloat, SwooshL,
                                                 SwooshR, Whiten, balancer_swoosh, limit_param_value)


class ConvNeXt(nn.Module):
    """depthwise 7x7 -> pointwise (x3) -> SwooshL -> pointwise, plus bypass (reference :26-132).
    Works on channel-last (N,H,W,C) activations: the pointwise convs are plain GEMMs over the
    last dim, the depthwise conv is the HIP NHWC stencil; parameters keep nn.Conv2d shapes."""

    def __init__(self, channels: int, hidden_ratio: int = 3, kernel_size: Tuple[int, int] = (7, 7),
                 layerdrop_rate: FloatLike = None):
        super().__init__()
        self.padding = ((kernel_size[0] - 1) // 2, (kernel_size[1] - 1) // 2)
        hidden = channels * hidden_ratio
        if layerdrop_rate is None:
            layerdrop_rate = ScheduledFloat((0.0, 0.2), (20000.0, 0.015))
        self.layerdrop_rate = layerdrop_rate
        self.depthwise_conv = nn.Conv2d(channels, channels, groups=channels,
                                        kernel_size=kernel_size, padding=self.padding)
        self.pointwise_conv1 = nn.Conv2d(channels, hidden, kernel_size=1)
        self.hidden_balancer = Balancer(hidden, channel_dim=-1, min_positive=0.3, max_positive=1.0,
                                        min_abs=0.75, max_abs=5.0)
        self.activation = SwooshL()
        self.pointwise_conv2 = ScaledConv2d(hidden, channels, kernel_size=1, initial_scale=0.01)
        self.out_balancer = Balancer(channels, channel_dim=-1, min_positive=0.4, max_positive=0.6,
                                     min_abs=1.0, max_abs=6.0)
        self.out_whiten = Whiten(num_groups=1, whitening_limit=5.0, prob=(0.025, 0.25),
                                 grad_scale=0.01)

    def forward(self, x: Tensor) -> Tensor:
        """x: (N,H,W,C) channel-last."""
        mask = None
        if self.training:
            rate = float(self.layerdrop_rate)
            if rate != 0.0:
                mask = rng.rand(x.shape[0], 1, 1, 1, dtype=x.dtype, device=x.device) > rate
        if mask is None and x.is_cuda and tuple(self.depthwise_conv.kernel_size) == (7, 7):
            # the residual add rides in the second pointwise GEMM's epilogue, its gradient in the
            # depthwise backward-data pass: no add pass over the (N,T,F,C) map in either direction
            x, bypass = zk.dwconv2d_nhwc_tap(x, self.depthwise_conv.weight, self.depthwise_conv.bias)
            # pointwise -> [Balancer] -> SwooshL -> pointwise + bypass as ONE node: the activation and
            # its derivative ride in the GEMMs' epilogues (zk.ffn_block); the Balancer's random
            # draw is taken here, where the module would take it
            hb = self.hidden_balancer
            x = zk.ffn_block(x, self.pointwise_conv1.weight, self.pointwise_conv1.bias,
                             self.pointwise_conv2.weight, self.pointwise_conv2.bias, bypass,
                             hb.cfg(2) if hb.fires(x) else None)
        else:
            bypass = x
            x = zk.dwconv2d_nhwc(x, self.depthwise_conv.weight, self.depthwise_conv.bias)
            x = zk.linear_big_m(x, self.pointwise_conv1.weight.flatten(1), self.pointwise_conv1.bias)
            x = self.hidden_balancer(x)
            x = self.activation(x)
            x = zk.linear_big_m(x, self.pointwise_conv2.weight.flatten(1), self.pointwise_conv2.bias)
            if mask is not None:
                x = x * mask
            x = bypass + x
        x = self.out_balancer(x)
        if x.requires_grad:
            x = self.out_whiten(x)
        return x


class Conv2dSubsampling(nn.Module):
    """(N,T,idim) -> (N,(T-7)//2,odim)   (reference :181-319).  Same parameters as the
    reference (nn.Conv2d weights in (Cout,Cin,kh,kw)), computed channel-last: the 3x3 convs are
    im2col + GEMM, everything elementwise streams over (N,T,F,C)."""

    def __init__(self, in_channels: int, out_channels: int, layer1_channels: int = 8,
                 layer2_channels: int = 32, layer3_channels: int = 128,
                 dropout: FloatLike = 0.1) -> None:
        assert in_channels >= 7
        super().__init__()
        self.conv = nn.Sequential(
            nn.Conv2d(1, layer1_channels, kernel_size=3, padding=(0, 1)),
            ScaleGrad(0.2),
            Balancer(layer1_channels, channel_dim=-1, max_abs=1.0),
            SwooshR(),
            nn.Conv2d(layer1_channels, layer2_channels, kernel_size=3, stride=2, padding=0),
            Balancer(layer2_channels, channel_dim=-1, max_abs=4.0),
            SwooshR(),
            nn.Conv2d(layer2_channels, layer3_channels, kernel_size=3, stride=(1, 2)),
            Balancer(layer3_channels, channel_dim=-1, max_abs=4.0),
            SwooshR(),
        )
        self.convnext = ConvNeXt(layer3_channels, kernel_size=(7, 7))
        self.out_width = (((in_channels - 1) // 2) - 1) // 2
        self.layer3_channels = layer3_channels
        self.out = Linear(self.out_width * layer3_channels, out_channels)
        self.out_whiten = Whiten(num_groups=1,
                                 whitening_limit=ScheduledFloat((0.0, 4.0), (20000.0, 8.0),
                                                                default=4.0),
                                 prob=(0.025, 0.25), grad_scale=0.02)
        self.out_norm = BiasNorm(out_channels)
        self.dropout = Dropout3(dropout, shared_dim=1)

    def forward(self, x: Tensor, x_lens: Tensor) -> Tuple[Tensor, Tensor]:
        x = x.unsqueeze(-1)                                     # (N,T,F,1) channel-last
        mods = list(self.conv)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Conv2d):
                gs = None
                if (i + 1 < len(mods) and isinstance(mods[i + 1], ScaleGrad)
                        and zk.conv3x3_direct_c1(x, m.weight, m.stride, m.padding[1])):
                    gs = mods[i + 1].alpha if self.training else None   # ScaleGrad is identity forward
                    i += 1
                x = zk.conv3x3_nhwc(x, m.weight, m.bias, m.stride, pad_w=m.padding[1],  # freq axis
                                    grad_scale=gs)
            elif (isinstance(m, Balancer) and i + 1 < len(mods) and isinstance(mods[i + 1], SwooshR)
                  and x.is_cuda and m.channel_dim in (-1, x.ndim - 1)):
                # Balancer + SwooshR: one autograd node, one backward pass over these (large) maps
                x = balancer_swoosh(m, x, False)
                i += 1
            else:
                x = m(x)
            i += 1
        x = self.convnext(x)                                    # (N,T',F',C)
        b, t, f, c = x.shape
        # reference flattens (c,f) c-major: out.weight columns are indexed c*F' + f
        x = zk.linear_col_perm(x.reshape(b, t, f * c), self.out.weight, self.out.bias, f, c)
        x = self.out_whiten(x)
        nm = self.out_norm
        if x.is_cuda and x.dim() == 3:
            # stored time-major: the encoder's x.transpose(0, 1) right after this module is then a
            # contiguous (T,B,C) tensor without a copy (and so is the gradient on the way back)
            ls = limit_param_value(nm.log_scale, min=float(nm.log_scale_min), max=float(nm.log_scale_max),
                                   training=nm.training)
            x = zk.bias_norm_time_major(x, nm.bias, ls)
        else:
            x = nm(x)
        x = self.dropout(x)
        x_lens = (x_lens - 7) // 2
        return x, x_lens
