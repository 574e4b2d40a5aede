"""Zipformer2 encoder for the MI355X training path.

Host-side mirror of the reference's model/encoder/zipformer.py: same config dataclass, class
names, constructor arguments and state_dict keys (checkpoints interchange), same forward
contract `forward(x[B,T,F], x_lens[B]) -> (out[B,T',D], lens)`.  The computation is arranged
for the GPU: activations stay time-major (T,B,C) end to end (the reference permutes to
(B,C,T) around each conv module), relative-position scores are produced with the rel->abs
shift folded into index arithmetic, and every hot op goes through speech2text_amd.zip_kernels
(the seam to the HIP C ABI).  Streaming / ONNX-export entry points of the reference are out of
scope (training hot path only).
"""
import copy
import dataclasses
import math
import random
from typing import List, Optional, Tuple, Union

import torch
from torch import Tensor, nn

from speech2text_amd import rng
from speech2text_amd import zip_kernels as zk
from speech2text_amd import zip_layer as zl
from speech2text_amd.model.functions.masking import make_pad_mask
from speech2text_amd.model.layer.scaling import (Linear, ActivationDropoutAndLinear, Balancer, BiasNorm,
                                                 ChunkCausalDepthwiseConv1d, Dropout2, FloatLike,
                                                 Identity, ScaledLinear, ScheduledFloat, Whiten,
                                                 convert_num_channels, limit_param_value,
                                                 penalize_abs_values_gt, _rand)
from speech2text_amd.model.layer.subsampling import Conv2dSubsampling


@dataclasses.dataclass
class Zipformer2Config:
    """Same keys/defaults as the reference's Zipformer2Config (zipformer.py:48-71)."""
    feature_dim: int = 80
    output_downsampling_factor: int = 2
    downsampling_factor: Tuple[int] = (2, 4)
    encoder_dim: Union[int, Tuple[int]] = 384
    num_encoder_layers: Union[int, Tuple[int]] = 4
    encoder_unmasked_dim: Union[int, Tuple[int]] = 256
    query_head_dim: Union[int, Tuple[int]] = 24
    pos_head_dim: Union[int, Tuple[int]] = 4
    value_head_dim: Union[int, Tuple[int]] = 12
    num_heads: Union[int, Tuple[int]] = 8
    feedforward_dim: Union[int, Tuple[int]] = 1536
    cnn_module_kernel: Union[int, Tuple[int]] = 31
    pos_dim: int = 192
    dropout: FloatLike = None
    warmup_batches: float = 4000.0
    causal: bool = False
    chunk_size: Tuple[int] = (-1,)
    left_context_frames: Tuple[int] = (-1,)
    for_ctc: bool = False
    num_tokens: int = 1000


def _whitening_schedule(x: float, ratio: float = 2.0) -> ScheduledFloat:
    return ScheduledFloat((0.0, x), (20000.0, ratio * x), default=x)


def _per_stack(x, n):
    if isinstance(x, int):
        x = (x,)
    x = tuple(x)
    if len(x) == 1:
        x = x * n
    assert len(x) == n, (x, n)
    return x


class Zipformer2(nn.Module):
    def __init__(self, config: Zipformer2Config) -> None:
        super().__init__()
        n = len(config.downsampling_factor)
        dropout = config.dropout
        if dropout is None:
            dropout = ScheduledFloat((0.0, 0.3), (20000.0, 0.1))
        self._feature_dim = config.feature_dim
        self.output_downsampling_factor = config.output_downsampling_factor
        self.downsampling_factor = tuple(config.downsampling_factor)
        self.encoder_dim = _per_stack(config.encoder_dim, n)
        self.encoder_unmasked_dim = _per_stack(config.encoder_unmasked_dim, n)
        self.num_encoder_layers = _per_stack(config.num_encoder_layers, n)
        self.query_head_dim = _per_stack(config.query_head_dim, n)
        self.value_head_dim = _per_stack(config.value_head_dim, n)
        self.pos_head_dim = _per_stack(config.pos_head_dim, n)
        self.num_heads = _per_stack(config.num_heads, n)
        self.feedforward_dim = _per_stack(config.feedforward_dim, n)
        self.cnn_module_kernel = _per_stack(config.cnn_module_kernel, n)
        self.causal = config.causal
        self.chunk_size = tuple(config.chunk_size) if not isinstance(config.chunk_size, int) \
            else (config.chunk_size,)
        self.left_context_frames = tuple(config.left_context_frames) \
            if not isinstance(config.left_context_frames, int) else (config.left_context_frames,)
        for u, d in zip(self.encoder_unmasked_dim, self.encoder_dim):
            assert u <= d

        self._encoder_embed = Conv2dSubsampling(
            in_channels=config.feature_dim, out_channels=self.encoder_dim[0],
            dropout=ScheduledFloat((0.0, 0.3), (20000.0, 0.1)))

        stacks = []
        for i in range(n):
            layer = Zipformer2EncoderLayer(
                embed_dim=self.encoder_dim[i], pos_dim=config.pos_dim,
                num_heads=self.num_heads[i], query_head_dim=self.query_head_dim[i],
                pos_head_dim=self.pos_head_dim[i], value_head_dim=self.value_head_dim[i],
                feedforward_dim=self.feedforward_dim[i], dropout=dropout,
                cnn_module_kernel=self.cnn_module_kernel[i], causal=config.causal)
            enc = Zipformer2Encoder(
                layer, self.num_encoder_layers[i], pos_dim=config.pos_dim, dropout=dropout,
                warmup_begin=config.warmup_batches * (i + 1) / (n + 1),
                warmup_end=config.warmup_batches * (i + 2) / (n + 1),
                final_layerdrop_rate=0.035 * (self.downsampling_factor[i] ** 0.5))
            if self.downsampling_factor[i] != 1:
                enc = DownsampledZipformer2Encoder(enc, dim=self.encoder_dim[i],
                                                   downsample=self.downsampling_factor[i],
                                                   dropout=dropout)
            stacks.append(enc)
        self.encoders = nn.ModuleList(stacks)
        self.downsample_output = SimpleDownsample(max(self.encoder_dim),
                                                  downsample=config.output_downsampling_factor,
                                                  dropout=dropout)
        self._for_ctc = config.for_ctc
        self._ctc_projection = Linear(max(self.encoder_dim), config.num_tokens) \
            if config.for_ctc else nn.Identity()

    # -- randomness that is drawn once per forward (reference zipformer.py:229-317)
    def get_feature_masks(self, x: Tensor):
        n = len(self.encoder_dim)
        if not self.training:
            return [1.0] * n
        _, B, d0 = x.shape
        assert d0 == self.encoder_dim[0]
        p = 0.125
        m1 = (rng.rand(1, B, 1, device=x.device) > p).to(x.dtype)
        m2 = torch.logical_and(m1, (rng.rand(1, B, 1, device=x.device) > p).to(x.dtype))
        # channel -> {1, m1, m2}: one gather per stack from (ones | m1 | m2) through a cached channel
        # selector (was ones + two sliced multiplies per stack: 30 launches of ~5 us)
        m = torch.cat((torch.ones_like(m1), m1, m2.to(x.dtype)), dim=-1)
        sel = self._fm_selectors(x.device)
        return [m.index_select(-1, sel[i]) for i in range(n)]

    def _fm_selectors(self, device):
        cache = self.__dict__.setdefault("_fm_sel", {})
        sel = cache.get(device)
        if sel is None:
            sel = []
            for c, u1 in zip(self.encoder_dim, self.encoder_unmasked_dim):
                u2 = u1 + (c - u1) // 2
                idx = torch.zeros(c, dtype=torch.long)
                idx[u1:u2] = 1
                idx[u2:] = 2
                sel.append(idx.to(device))
            cache[device] = sel
        return sel

    def get_chunk_info(self) -> Tuple[int, int]:
        if not self.causal:
            return -1, -1
        chunk_size = random.choice(self.chunk_size)
        if chunk_size == -1:
            return -1, -1
        left = random.choice(self.left_context_frames) // chunk_size
        return chunk_size, (1 if left == 0 else left)

    def _get_attn_mask(self, x: Tensor, chunk_size: int, left_context_chunks: int):
        if chunk_size <= 0:
            return None
        assert all(chunk_size % d == 0 for d in self.downsampling_factor)
        if left_context_chunks >= 0:
            assert all(chunk_size * left_context_chunks >= (k // 2) * d
                       for k, d in zip(self.cnn_module_kernel, self.downsampling_factor))
        else:
            left_context_chunks = 1000000
        c = torch.arange(x.shape[0], dtype=torch.int32, device=x.device) // chunk_size
        return torch.logical_or(c.unsqueeze(0) > c.unsqueeze(1),
                                c.unsqueeze(0) < c.unsqueeze(1) - left_context_chunks)

    def forward(self, x: Tensor, x_lens: Tensor) -> Tuple[Tensor, Tensor]:
        x, x_lens = self._encoder_embed(x, x_lens)
        pad_mask = make_pad_mask(x_lens, x.shape[1])
        x = x.transpose(0, 1)                       # (T,B,C), kept for the whole encoder
        feature_masks = self.get_feature_masks(x)
        chunk_size, left_context_chunks = self.get_chunk_info()
        attn_mask = self._get_attn_mask(x, chunk_size, left_context_chunks)
        outputs = []
        for i, stack in enumerate(self.encoders):
            ds = self.downsampling_factor[i]
            x = convert_num_channels(x, self.encoder_dim[i])
            x = stack(x, chunk_size=chunk_size, feature_mask=feature_masks[i],
                      src_key_padding_mask=pad_mask[..., ::ds], attn_mask=attn_mask)
            outputs.append(x)
        x = self._get_full_dim_output(outputs)
        x = self.downsample_output(x, batch_major=True)   # (stored (B,T,C): the transpose below is free)
        assert self.output_downsampling_factor == 2
        lengths = (x_lens + 1) // 2
        x = x.transpose(0, 1)
        if self._for_ctc:
            x = self._ctc_projection(x)
        return x, lengths

    # -- streaming inference (reference :391-406, :529-663); speech2text_amd/model/encoder/
    #    zipformer_streaming.py
    def get_init_states(self, batch_size: int = 1, device=torch.device("cpu")) -> List[Tensor]:
        from speech2text_amd.model.encoder import zipformer_streaming as zs
        return zs.get_init_states(self, batch_size, device)

    def streaming_step(self, x: Tensor, states: List[Tensor]) -> Tuple[Tensor, List[Tensor]]:
        from speech2text_amd.model.encoder import zipformer_streaming as zs
        return zs.streaming_step(self, x, states)

    def streaming_forward(self, x: Tensor, x_lens: Tensor, chunk_size=(32,),
                          left_context_frames=(128,)) -> Tuple[Tensor, Tensor]:
        from speech2text_amd.model.encoder import zipformer_streaming as zs
        return zs.simulated_streaming_forward(self, x, x_lens, chunk_size, left_context_frames)

    def _get_full_dim_output(self, outputs: List[Tensor]):
        pieces = [outputs[-1]]
        cur = self.encoder_dim[-1]
        for i in range(len(self.encoder_dim) - 2, -1, -1):
            d = self.encoder_dim[i]
            if d > cur:
                pieces.append(outputs[i][..., cur:d])
                cur = d
        assert cur == max(self.encoder_dim)
        return torch.cat(pieces, dim=-1) if len(pieces) > 1 else pieces[0]


class BypassModule(nn.Module):
    """src_orig + (src - src_orig) * clamp-limited learnable scale (reference :1499-1555)."""

    def __init__(self, embed_dim: int, skip_rate: FloatLike = 0.0,
                 straight_through_rate: FloatLike = 0.0,
                 scale_min: FloatLike = None, scale_max: FloatLike = 1.0):
        super().__init__()
        self.bypass_scale = nn.Parameter(torch.full((embed_dim,), 0.5))
        self.skip_rate = copy.deepcopy(skip_rate)
        self.straight_through_rate = copy.deepcopy(straight_through_rate)
        if scale_min is None:
            scale_min = ScheduledFloat((0.0, 0.9), (20000.0, 0.2), default=0)
        self.scale_min = copy.deepcopy(scale_min)
        self.scale_max = copy.deepcopy(scale_max)

    def _get_bypass_scale(self, batch_size: int):
        if not self.training:
            return self.bypass_scale
        ans = limit_param_value(self.bypass_scale, min=float(self.scale_min),
                                max=float(self.scale_max))
        skip_rate = float(self.skip_rate)
        if skip_rate != 0.0:
            ans = ans * (rng.rand(batch_size, 1, device=ans.device) > skip_rate)
        st = float(self.straight_through_rate)
        if st != 0.0:
            mask = rng.rand(batch_size, 1, device=ans.device) < st
            ans = torch.maximum(ans, mask.to(ans.dtype))
        return ans

    def forward(self, src_orig: Tensor, src: Tensor):
        return zk.bypass_combine(src_orig, src, self._get_bypass_scale(src.shape[1]))


class SimpleDownsample(nn.Module):
    """softmax(bias)-weighted sum of `downsample` consecutive frames (reference :1653-1695)."""

    def __init__(self, channels: int, downsample: int, dropout: FloatLike):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(downsample))
        self.name = None
        self.dropout = copy.deepcopy(dropout)
        self.downsample = downsample

    def forward(self, src: Tensor, batch_major: bool = False) -> Tensor:
        return zk.simple_downsample(src, self.bias, self.downsample, batch_major)


class SimpleUpsample(nn.Module):
    def __init__(self, num_channels: int, upsample: int):
        super().__init__()
        self.upsample = upsample

    def forward(self, src: Tensor, out_len: Optional[int] = None) -> Tensor:
        if out_len is None:
            out_len = src.shape[0] * self.upsample
        return zk.simple_upsample(src, self.upsample, out_len)


class CompactRelPositionalEncoding(nn.Module):
    """Relative position table pe[2T-1, D]: offsets compressed by log then atan, expanded in a
    Fourier basis, last column 1 (reference :1722-1833).  Cached per (T, device)."""

    def __init__(self, embed_dim: int, dropout_rate: FloatLike, max_len: int = 1000,
                 length_factor: float = 1.0) -> None:
        super().__init__()
        assert embed_dim % 2 == 0 and length_factor >= 1.0
        self.embed_dim = embed_dim
        self.dropout = Dropout2(dropout_rate)
        self.length_factor = length_factor
        self._cache = {}

    def table(self, T: int, device) -> Tensor:
        key = (T, str(device))
        pe = self._cache.get(key)
        if pe is None:
            D = self.embed_dim
            x = torch.arange(-(T - 1), T, device=device).to(torch.float32).unsqueeze(1)
            freqs = 1 + torch.arange(D // 2, device=device)
            cl = D ** 0.5
            xc = cl * x.sign() * ((x.abs() + cl).log() - math.log(cl))
            ls = self.length_factor * D / (2.0 * math.pi)
            xa = (xc / ls).atan()
            pe = torch.zeros(x.shape[0], D, device=device)
            pe[:, 0::2] = (xa * freqs).cos()
            pe[:, 1::2] = (xa * freqs).sin()
            pe[:, -1] = 1.0
            if len(self._cache) > 16:
                self._cache.clear()
            self._cache[key] = pe
        return pe

    def forward(self, x: Tensor, left_context_len: int = 0) -> Tensor:
        """(1, left_context_len + 2T-1, D): offsets -(T+left-1) .. T-1 (reference :1815-1833)."""
        T = x.size(0)
        pe = self.table(T + left_context_len, x.device)
        if left_context_len:
            pe = pe[:left_context_len + 2 * T - 1]
        return self.dropout(pe.unsqueeze(0))


class Zipformer2Encoder(nn.Module):
    def __init__(self, encoder_layer: nn.Module, num_layers: int, pos_dim: int, dropout: float,
                 warmup_begin: float, warmup_end: float, initial_layerdrop_rate: float = 0.5,
                 final_layerdrop_rate: float = 0.05) -> None:
        super().__init__()
        self.encoder_pos = CompactRelPositionalEncoding(pos_dim, dropout_rate=0.15,
                                                        length_factor=1.0)
        self.layers = nn.ModuleList([copy.deepcopy(encoder_layer) for _ in range(num_layers)])
        self.num_layers = num_layers
        assert 0 <= warmup_begin <= warmup_end
        delta = (1.0 / num_layers) * (warmup_end - warmup_begin)
        cur = warmup_begin
        for i in range(num_layers):
            self.layers[i].bypass.skip_rate = ScheduledFloat(
                (cur, initial_layerdrop_rate), (cur + delta, final_layerdrop_rate), default=0.0)
            cur += delta

    def forward(self, src: Tensor, chunk_size: int = -1, feature_mask=1.0,
                attn_mask: Optional[Tensor] = None,
                src_key_padding_mask: Optional[Tensor] = None) -> Tensor:
        pos_emb = self.encoder_pos(src)
        masked = isinstance(feature_mask, Tensor)
        out = src * feature_mask if masked else src
        for layer in self.layers:
            out = layer(out, pos_emb, chunk_size=chunk_size, attn_mask=attn_mask,
                        src_key_padding_mask=src_key_padding_mask,
                        feature_mask=feature_mask if masked else None)
        return out


class DownsampledZipformer2Encoder(nn.Module):
    def __init__(self, encoder: nn.Module, dim: int, downsample: int, dropout: FloatLike):
        super().__init__()
        self.downsample_factor = downsample
        self.downsample = SimpleDownsample(dim, downsample, dropout)
        self.num_layers = encoder.num_layers
        self.encoder = encoder
        self.upsample = SimpleUpsample(dim, downsample)
        self.out_combiner = BypassModule(dim, straight_through_rate=0)

    def forward(self, src: Tensor, chunk_size: int = -1, feature_mask=1.0,
                attn_mask: Optional[Tensor] = None,
                src_key_padding_mask: Optional[Tensor] = None) -> Tensor:
        ds = self.downsample_factor
        orig = src
        src = self.downsample(src)
        if attn_mask is not None:
            attn_mask = attn_mask[::ds, ::ds]
        src = self.encoder(src, chunk_size=chunk_size // ds, feature_mask=feature_mask,
                           attn_mask=attn_mask, src_key_padding_mask=src_key_padding_mask)
        # upsample + out_combiner in one pass (the upsampled tensor is never materialised); the
        # combiner's limit_param_value draw happens here, as in its forward
        scale = self.out_combiner._get_bypass_scale(src.shape[1])
        return zk.bypass_upsampled(orig, src, scale, self.upsample.upsample)


class RelPositionMultiheadAttentionWeights(nn.Module):
    """softmax(q.k + p.pos_rel) with -1000 masking -> (H,B,T,T)  (reference :1836-2077)."""

    def __init__(self, embed_dim: int, pos_dim: int, num_heads: int, query_head_dim: int,
                 pos_head_dim: int, dropout: float = 0.0,
                 pos_emb_skip_rate: FloatLike = None) -> None:
        super().__init__()
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.query_head_dim = query_head_dim
        self.pos_head_dim = pos_head_dim
        self.dropout = dropout
        if pos_emb_skip_rate is None:
            pos_emb_skip_rate = ScheduledFloat((0.0, 0.5), (4000.0, 0.0))
        self.pos_emb_skip_rate = copy.deepcopy(pos_emb_skip_rate)
        self.name = None
        in_proj_dim = (2 * query_head_dim + pos_head_dim) * num_heads
        self.in_proj = ScaledLinear(embed_dim, in_proj_dim, bias=True,
                                    initial_scale=query_head_dim ** -0.25)
        self.whiten_keys = Whiten(num_groups=num_heads, whitening_limit=_whitening_schedule(3.0),
                                  prob=(0.025, 0.25), grad_scale=0.025)
        self.balance_keys = Balancer(query_head_dim * num_heads, channel_dim=-1, min_positive=0.4,
                                     max_positive=0.6, min_abs=0.0, max_abs=100.0, prob=0.025)
        self.linear_pos = ScaledLinear(pos_dim, num_heads * pos_head_dim, bias=False,
                                       initial_scale=0.05)
        self._next_shared = None

    def forward(self, x: Tensor, pos_emb: Tensor, key_padding_mask: Optional[Tensor] = None,
                attn_mask: Optional[Tensor] = None) -> Tensor:
        H, qd, pd = self.num_heads, self.query_head_dim, self.pos_head_dim
        self.last_deferrable = False
        qkp = self.in_proj(x)
        T, B, _ = qkp.shape
        # gradient shaping of the key slice (identity in forward): only when one of them fires
        # is the slice cut out and stitched back
        k = qkp[..., H * qd:2 * H * qd]
        fb = self.balance_keys.fires(k)
        fw = self.whiten_keys.fires(k)
        if fb or fw:
            if fb:
                k = self.balance_keys.shape_grad(k)
            if fw:
                k = self.whiten_keys.shape_grad(k)
            qkp = torch.cat((qkp[..., :H * qd], k, qkp[..., 2 * H * qd:]), dim=-1)
        pos_proj = None
        if not self.training or _rand() >= float(self.pos_emb_skip_rate):
            pos_proj = self.linear_pos(pos_emb).reshape(2 * T - 1, H * pd)
        penalize = None
        if self.training and _rand() < 0.1:
            penalize = lambda s: penalize_abs_values_gt(s, limit=25.0, penalty=1.0e-04,  # noqa
                                                        name=self.name)
        if penalize is None and self.dropout == 0.0:
            # the consumers' gradients w.r.t. W can be contracted inside our backward
            self._next_shared = zk.AttnShared()
            self._next_shared.device = qkp.device
            self.last_deferrable = True
            self.last_shared = self._next_shared
        w = zk.relpos_attention_weights(qkp, pos_proj, H, qd, pd, attn_mask, key_padding_mask,
                                        penalize, shared=self._next_shared)
        self._next_shared = None
        if self.dropout != 0.0:
            w = nn.functional.dropout(w, p=self.dropout, training=self.training)
        return w


class SelfAttention(nn.Module):
    def __init__(self, embed_dim: int, num_heads: int, value_head_dim: int) -> None:
        super().__init__()
        self.in_proj = Linear(embed_dim, num_heads * value_head_dim, bias=True)
        self.out_proj = ScaledLinear(num_heads * value_head_dim, embed_dim, bias=True,
                                     initial_scale=0.05)
        self.whiten = Whiten(num_groups=1, whitening_limit=_whitening_schedule(7.5, ratio=3.0),
                             prob=(0.025, 0.25), grad_scale=0.01)

    def forward(self, x: Tensor, attn_weights: Tensor, residual: Optional[Tensor] = None) -> Tensor:
        """With `residual` the result is residual + module(x); when the output Whiten does not
        fire this step the add rides in the out_proj GEMM's epilogue."""
        if residual is x:
            v, residual = zk.linear_pass(x, self.in_proj.weight, self.in_proj.bias)
        else:
            v = self.in_proj(x)
        x = zk.attention_apply(attn_weights, v, attn_weights.shape[0])
        fw = self.whiten.fires(x)
        if residual is not None and not fw:
            return zk.linear(x, self.out_proj.weight, self.out_proj.bias, residual=residual)
        x = self.out_proj(x)
        if fw:
            x = self.whiten.shape_grad(x)
        return x if residual is None else residual + x


class FeedforwardModule(nn.Module):
    def __init__(self, embed_dim: int, feedforward_dim: int, dropout: FloatLike):
        super().__init__()
        self.in_proj = Linear(embed_dim, feedforward_dim)
        self.hidden_balancer = Balancer(feedforward_dim, channel_dim=-1, min_positive=0.3,
                                        max_positive=1.0, min_abs=0.75, max_abs=5.0)
        self.out_proj = ActivationDropoutAndLinear(feedforward_dim, embed_dim,
                                                   activation="SwooshL", dropout_p=dropout,
                                                   dropout_shared_dim=0, bias=True,
                                                   initial_scale=0.1)
        self.out_whiten = Whiten(num_groups=1, whitening_limit=_whitening_schedule(7.5),
                                 prob=(0.025, 0.25), grad_scale=0.01)

    def forward(self, x: Tensor, residual: Optional[Tensor] = None, post=None):
        """`post`: the layer's Balancer applied to this module's output (its random draw is made
        here, right after out_whiten's, i.e. in the reference's order).  With `residual` the
        result is residual + post(module(x)); if neither gradient-shaping op fires this step the
        add is fused into the out_proj GEMM."""
        if residual is x:
            h, residual = zk.linear_pass(x, self.in_proj.weight, self.in_proj.bias)
        else:
            h = self.in_proj(x)
        x = self.hidden_balancer(h)
        fw = self.out_whiten.fires(x)
        fp = post.fires(x) if post is not None else False
        if residual is not None and not fw and not fp:
            return self.out_proj(x, residual=residual)
        x = self.out_proj(x)
        if fw:
            x = self.out_whiten.shape_grad(x)
        if fp:
            x = post.shape_grad(x)
        return x if residual is None else residual + x


class NonlinAttention(nn.Module):
    """(x * tanh(s)) attended with head-0 weights, times y, projected (reference :2381-2483)."""

    def __init__(self, channels: int, hidden_channels: int) -> None:
        super().__init__()
        self.hidden_channels = hidden_channels
        self.in_proj = Linear(channels, hidden_channels * 3, bias=True)
        self.balancer = Balancer(hidden_channels, channel_dim=-1,
                                 min_positive=ScheduledFloat((0.0, 0.25), (20000.0, 0.05)),
                                 max_positive=ScheduledFloat((0.0, 0.75), (20000.0, 0.95)),
                                 min_abs=0.5, max_abs=5.0)
        self.tanh = nn.Tanh()
        self.out_proj = ScaledLinear(hidden_channels, channels, bias=True, initial_scale=0.05)
        self.whiten1 = Whiten(num_groups=1, whitening_limit=_whitening_schedule(5.0),
                              prob=(0.025, 0.25), grad_scale=0.01)
        self.whiten2 = Whiten(num_groups=1, whitening_limit=_whitening_schedule(5.0, ratio=3.0),
                              prob=(0.025, 0.25), grad_scale=0.01)

    def forward(self, x: Tensor, attn_weights: Tensor, residual: Optional[Tensor] = None,
                post=None) -> Tensor:
        if residual is x:
            u, residual = zk.linear_pass(x, self.in_proj.weight, self.in_proj.bias)
        else:
            u = self.in_proj(x)                   # (T,B,3C) = [s | x | y]
        if attn_weights.shape[0] == 1 and u.shape[-1] % 3 == 0:
            # fused core; the random draws keep the reference's order (balancer, then whiten1)
            fb = self.balancer.fires(u)
            fw1 = self.whiten1.fires(u)
            x = zk.nonlin_core(u, attn_weights, self.balancer.cfg(3) if fb else None,
                               self.whiten1 if fw1 else None)
        else:
            s, x, y = u.chunk(3, dim=2)
            s = self.tanh(self.balancer(s))
            x = self.whiten1(x) * s
            x = zk.attention_apply(attn_weights, x, attn_weights.shape[0])
            x = x * y
        fw = self.whiten2.fires(x)
        fp = post.fires(x) if post is not None else False
        if residual is not None and not fw and not fp:
            return zk.linear(x, self.out_proj.weight, self.out_proj.bias, residual=residual)
        x = self.out_proj(x)
        if fw:
            x = self.whiten2.shape_grad(x)
        if fp:
            x = post.shape_grad(x)
        return x if residual is None else residual + x


class ConvolutionModule(nn.Module):
    """in_proj -> x*sigmoid(gate) -> zero padding -> (chunk-causal) depthwise conv -> SwooshR
    -> out_proj, all time-major (reference :2547-2695)."""

    def __init__(self, channels: int, kernel_size: int, causal: bool) -> None:
        super().__init__()
        assert (kernel_size - 1) % 2 == 0
        self.causal = causal
        self.in_proj = Linear(channels, 2 * channels)
        self.balancer1 = Balancer(channels, channel_dim=-1,
                                  min_positive=ScheduledFloat((0.0, 0.05), (8000.0, 0.025)),
                                  max_positive=1.0, min_abs=1.5,
                                  max_abs=ScheduledFloat((0.0, 5.0), (8000.0, 10.0), default=1.0))
        self.depthwise_conv = (ChunkCausalDepthwiseConv1d(channels=channels,
                                                          kernel_size=kernel_size)
                               if causal else nn.Conv1d(channels, channels, groups=channels,
                                                        kernel_size=kernel_size,
                                                        padding=kernel_size // 2))
        # reference balancer2 acts on (B,C,T) with channel_dim=1; the data is the same set of
        # per-channel samples here, laid out (T,B,C)
        self.balancer2 = Balancer(channels, channel_dim=-1,
                                  min_positive=ScheduledFloat((0.0, 0.1), (8000.0, 0.05)),
                                  max_positive=1.0,
                                  min_abs=ScheduledFloat((0.0, 0.2), (20000.0, 0.5)),
                                  max_abs=10.0)
        self.whiten = Whiten(num_groups=1, whitening_limit=_whitening_schedule(7.5),
                             prob=(0.025, 0.25), grad_scale=0.01)
        self.out_proj = ActivationDropoutAndLinear(channels, channels, activation="SwooshR",
                                                   dropout_p=0.0, initial_scale=0.05)

    def forward(self, x: Tensor, src_key_padding_mask: Optional[Tensor] = None,
                chunk_size: int = -1, residual: Optional[Tensor] = None) -> Tensor:
        if residual is x:
            u, residual = zk.linear_pass(x, self.in_proj.weight, self.in_proj.bias)
        else:
            u = self.in_proj(x)                  # (T,B,2C): [x | gate pre-activation]
        C = u.shape[-1] // 2
        if self.balancer1.fires(u):
            # gradient shaping of the gate half only (identity in forward)
            u = torch.cat((u[..., :C], self.balancer1.shape_grad(u[..., C:])), dim=-1)
        if chunk_size >= 0:
            assert self.causal, "Must initialize model with causal=True if you use chunk_size"
        x = zk.glu_chunk_causal_dwconv(u, C, src_key_padding_mask, self.depthwise_conv,
                                       chunk_size)
        x = self.whiten(self.balancer2(x))
        return self.out_proj(x, residual=residual)     # out_proj is last: the add is always fused


class Zipformer2EncoderLayer(nn.Module):
    def __init__(self, embed_dim: int, pos_dim: int, num_heads: int, query_head_dim: int,
                 pos_head_dim: int, value_head_dim: int, feedforward_dim: int,
                 dropout: FloatLike = 0.1, cnn_module_kernel: int = 31, causal: bool = False,
                 attention_skip_rate: FloatLike = None, conv_skip_rate: FloatLike = None,
                 const_attention_rate: FloatLike = None, ff2_skip_rate: FloatLike = None,
                 ff3_skip_rate: FloatLike = None, bypass_skip_rate: FloatLike = None) -> None:
        super().__init__()
        SF = ScheduledFloat
        self.embed_dim = embed_dim
        if bypass_skip_rate is None:
            bypass_skip_rate = SF((0.0, 0.5), (4000.0, 0.02), default=0)
        self.bypass = BypassModule(embed_dim, skip_rate=bypass_skip_rate, straight_through_rate=0)
        self.bypass_mid = BypassModule(embed_dim, straight_through_rate=0)
        self.attention_skip_rate = copy.deepcopy(attention_skip_rate) if attention_skip_rate \
            is not None else SF((0.0, 0.2), (4000.0, 0.05), (16000, 0.0), default=0)
        self.conv_skip_rate = copy.deepcopy(conv_skip_rate) if conv_skip_rate is not None \
            else SF((0.0, 0.2), (4000.0, 0.05), (16000, 0.0), default=0)
        self.ff2_skip_rate = copy.deepcopy(ff2_skip_rate) if ff2_skip_rate is not None \
            else SF((0.0, 0.1), (4000.0, 0.01), (50000.0, 0.0))
        self.ff3_skip_rate = copy.deepcopy(ff3_skip_rate) if ff3_skip_rate is not None \
            else SF((0.0, 0.1), (4000.0, 0.01), (50000.0, 0.0))
        self.const_attention_rate = copy.deepcopy(const_attention_rate) \
            if const_attention_rate is not None else SF((0.0, 0.25), (4000.0, 0.025), default=0)

        self.self_attn_weights = RelPositionMultiheadAttentionWeights(
            embed_dim, pos_dim=pos_dim, num_heads=num_heads, query_head_dim=query_head_dim,
            pos_head_dim=pos_head_dim, dropout=0.0)
        self.self_attn1 = SelfAttention(embed_dim, num_heads, value_head_dim)
        self.self_attn2 = SelfAttention(embed_dim, num_heads, value_head_dim)
        self.feed_forward1 = FeedforwardModule(embed_dim, (feedforward_dim * 3) // 4, dropout)
        self.feed_forward2 = FeedforwardModule(embed_dim, feedforward_dim, dropout)
        self.feed_forward3 = FeedforwardModule(embed_dim, (feedforward_dim * 5) // 4, dropout)
        self.nonlin_attention = NonlinAttention(embed_dim, hidden_channels=3 * embed_dim // 4)
        self.conv_module1 = ConvolutionModule(embed_dim, cnn_module_kernel, causal=causal)
        self.conv_module2 = ConvolutionModule(embed_dim, cnn_module_kernel, causal=causal)
        self.bypass_scale = nn.Parameter(torch.full((embed_dim,), 0.5))   # unused, as upstream
        self.norm = BiasNorm(embed_dim)
        self.balancer1 = Balancer(embed_dim, channel_dim=-1, min_positive=0.45, max_positive=0.55,
                                  min_abs=0.2, max_abs=4.0)
        self.balancer_na = Balancer(embed_dim, channel_dim=-1, min_positive=0.3, max_positive=0.7,
                                    min_abs=SF((0.0, 0.004), (4000.0, 0.02)), prob=0.05)
        self.balancer_ff2 = Balancer(embed_dim, channel_dim=-1, min_positive=0.3,
                                     max_positive=0.7,
                                     min_abs=SF((0.0, 0.0), (4000.0, 0.1), default=0.0),
                                     max_abs=2.0, prob=0.05)
        self.balancer_ff3 = Balancer(embed_dim, channel_dim=-1, min_positive=0.3,
                                     max_positive=0.7,
                                     min_abs=SF((0.0, 0.0), (4000.0, 0.2), default=0.0),
                                     max_abs=4.0, prob=0.05)
        self.whiten = Whiten(num_groups=1, whitening_limit=_whitening_schedule(4.0, ratio=3.0),
                             prob=(0.025, 0.25), grad_scale=0.01)
        self.balancer2 = Balancer(embed_dim, channel_dim=-1, min_positive=0.45, max_positive=0.55,
                                  min_abs=0.1, max_abs=4.0)

    def _seq_mask(self, x: Tensor, rate: float) -> Optional[Tensor]:
        if rate == 0.0 or not self.training:
            return None
        return (rng.rand(x.shape[1], 1, device=x.device) > rate).to(x.dtype)

    def forward(self, src: Tensor, pos_emb: Tensor, chunk_size: int = -1,
                attn_mask: Optional[Tensor] = None,
                src_key_padding_mask: Optional[Tensor] = None,
                feature_mask: Optional[Tensor] = None) -> Tensor:
        """feature_mask: the encoder stack's (1,B,C) mask, applied to the output (reference
        Zipformer2Encoder.forward, zipformer.py:1095-1113, does it right after the layer call)."""
        if zl.eligible(self, src, attn_mask, src_key_padding_mask):
            # one autograd node for the whole layer (speech2text_amd/zip_layer.py)
            out = zl.run(self, src, pos_emb, chunk_size, attn_mask, src_key_padding_mask,
                         feature_mask)
            if out is not None:
                return out
        out = self._forward_modules(src, pos_emb, chunk_size, attn_mask, src_key_padding_mask)
        return out if feature_mask is None else out * feature_mask

    def _forward_modules(self, src: Tensor, pos_emb: Tensor, chunk_size: int,
                         attn_mask: Optional[Tensor],
                         src_key_padding_mask: Optional[Tensor]) -> Tensor:
        train = self.training
        src_orig = src
        attn_skip = float(self.attention_skip_rate) if train else 0.0
        w = self.self_attn_weights(src, pos_emb=pos_emb, attn_mask=attn_mask,
                                   key_padding_mask=src_key_padding_mask)
        src = self.feed_forward1(src, residual=src)
        amask = self._seq_mask(src, attn_skip)
        const_attn = train and _rand() < float(self.const_attention_rate)
        if self.self_attn_weights.last_deferrable and not const_attn:
            # one autograd edge into the weights; consumers hand their gradient factors over
            shared = self.self_attn_weights.last_shared
            w_na, w_a1, w_a2 = (zk.DeferredWeights(a, shared)
                                for a in zk._AttnFanout.apply(w, shared, 3))
            w0 = zk.attn_head0(w_na)
        else:
            w_a1 = w_a2 = w
            w0 = w[0:1]
            if const_attn:
                w0 = (w0 > 0.0).to(w0.dtype)
                w0 = w0 * (1.0 / w0.sum(dim=-1, keepdim=True))
        # the residual adds ride in each module's last GEMM (sequence masks: defaults are None)
        if amask is None:
            src = self.nonlin_attention(src, w0, residual=src, post=self.balancer_na)
            src = self.self_attn1(src, w_a1, residual=src)
        else:
            src = src + self.nonlin_attention(src, w0, post=self.balancer_na) * amask
            src = src + self.self_attn1(src, w_a1) * amask
        conv_skip = float(self.conv_skip_rate) if train else 0.0
        cm = None if conv_skip == 0.0 else 0
        if cm is None:
            src = self.conv_module1(src, chunk_size=chunk_size,
                                    src_key_padding_mask=src_key_padding_mask, residual=src)
        else:
            cv = self.conv_module1(src, chunk_size=chunk_size,
                                   src_key_padding_mask=src_key_padding_mask)
            src = src + cv * self._seq_mask(src, conv_skip)
        ff2_skip = float(self.ff2_skip_rate) if train else 0.0
        if ff2_skip == 0.0:
            src = self.feed_forward2(src, residual=src, post=self.balancer_ff2)
        else:
            ff = self.feed_forward2(src, post=self.balancer_ff2)
            src = src + ff * self._seq_mask(src, ff2_skip)
        src = self.bypass_mid(src_orig, src)
        if amask is None:
            src = self.self_attn2(src, w_a2, residual=src)
        else:
            src = src + self.self_attn2(src, w_a2) * amask
        if cm is None:
            src = self.conv_module2(src, chunk_size=chunk_size,
                                    src_key_padding_mask=src_key_padding_mask, residual=src)
        else:
            cv = self.conv_module2(src, chunk_size=chunk_size,
                                   src_key_padding_mask=src_key_padding_mask)
            src = src + cv * self._seq_mask(src, conv_skip)
        ff3_skip = float(self.ff3_skip_rate) if train else 0.0
        if ff3_skip == 0.0:
            src = self.feed_forward3(src, residual=src, post=self.balancer_ff3)
        else:
            ff = self.feed_forward3(src, post=self.balancer_ff3)
            src = src + ff * self._seq_mask(src, ff3_skip)
        src = self.norm(self.balancer1(src))
        src = self.bypass(src_orig, src)
        return self.whiten(self.balancer2(src))
