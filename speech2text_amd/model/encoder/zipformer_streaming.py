"""Chunk-by-chunk (streaming) inference of Zipformer2 on the MI355X kernels.

Mirrors the reference's streaming surface (model/encoder/zipformer.py): `get_init_states`
(:529-600), `streaming_step` (:601-663) and the per-module `streaming_forward`s it calls
(:465-527, :1223-1338, :1432-1496, :1616-1650, :2079-2190, :2282-2333, :2485-2541, :2697-2741;
model/layer/scaling.py:683-716; model/layer/subsampling.py:134-178, 321-391).  The state list
keeps the reference's order, shapes and layouts, so states are interchangeable.

Every dense op runs through the kernels the training path uses (no torch fallback): GEMMs,
Swoosh+GEMM, BiasNorm, bypass, down/upsampling and the NHWC convolutions as they are; the two
stateful pieces are mapped onto the existing kernels instead of getting slower special cases:

* attention with a cached left context of L frames is the square relative-position kernel over
  S = L + T positions whose first L query rows are empty (their weights are dropped): key j and
  query i see offset j - i in both formulations, and the table of S positions contains the
  reference's streaming slice -(S-1) .. T-1;
* the chunk-causal depthwise conv with a cached left pad is the fused conv kernel run over
  [zeros | cache | chunk] with chunk length T: the causal half reads back into the cache, the
  chunk-wise half stays inside the last chunk, and only the last chunk's output is kept.
"""
from typing import List, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

from speech2text_amd import zip_kernels as zk
from speech2text_amd.model.layer.scaling import convert_num_channels


# ------------------------------------------------------------------ states
def get_init_states(model, batch_size: int = 1, device=torch.device("cpu")) -> List[Tensor]:
    """For layer i, states[6i:6i+6] = (cached_key (L,B,H*qd), cached_nonlin_attn (1,B,L,3D/4),
    cached_val1, cached_val2 (L,B,H*vd), cached_conv1, cached_conv2 (B,D,K//2)); states[-2] the
    ConvNeXt left pad (B,C,3,F'); states[-1] processed_lens (B,) int64."""
    states = []
    left = model.left_context_frames[0]
    for i, stack in enumerate(model.encoders):
        D, H, ds = model.encoder_dim[i], model.num_heads[i], model.downsampling_factor[i]
        L = left // ds
        pad = model.cnn_module_kernel[i] // 2
        for _ in range(stack.num_layers):
            states += [torch.zeros(L, batch_size, H * model.query_head_dim[i], device=device),
                       torch.zeros(1, batch_size, L, 3 * D // 4, device=device),
                       torch.zeros(L, batch_size, H * model.value_head_dim[i], device=device),
                       torch.zeros(L, batch_size, H * model.value_head_dim[i], device=device),
                       torch.zeros(batch_size, D, pad, device=device),
                       torch.zeros(batch_size, D, pad, device=device)]
    emb = model._encoder_embed
    states.append(torch.zeros(batch_size, emb.layer3_channels, emb.convnext.padding[0],
                              emb.out_width, device=device))
    states.append(torch.zeros(batch_size, dtype=torch.int64, device=device))
    return states


# ------------------------------------------------------------------ modules
def _attn_weights(m, x: Tensor, pos_proj: Tensor, cached_key: Tensor, L: int, kpm: Tensor):
    """RelPositionMultiheadAttentionWeights.streaming_forward.  -> (W (H,B,T,L+T), new cache)."""
    H, qd, pd = m.num_heads, m.query_head_dim, m.pos_head_dim
    qkp = m.in_proj(x)
    T, B, W_ = qkp.shape
    full = torch.zeros(L + T, B, W_, dtype=qkp.dtype, device=qkp.device)
    full[L:] = qkp
    full[:L, :, H * qd:2 * H * qd] = cached_key
    new_key = full[T:, :, H * qd:2 * H * qd].contiguous()
    w = zk.relpos_attention_weights(full, pos_proj, H, qd, pd, None, kpm)
    return w[:, :, L:], new_key


def _self_attn(m, x: Tensor, w: Tensor, cached_val: Tensor, L: int):
    """SelfAttention.streaming_forward; the residual add rides in the out_proj GEMM."""
    T, B, _ = x.shape
    H = w.shape[0]
    v = torch.cat([cached_val, m.in_proj(x)], dim=0)
    new_val = v[T:]
    o = torch.matmul(w, v.reshape(L + T, B, H, -1).permute(2, 1, 0, 3))
    o = o.permute(2, 1, 0, 3).reshape(T, B, -1)
    return zk.linear(o, m.out_proj.weight, m.out_proj.bias, residual=x), new_val


def _nonlin_attention(m, x: Tensor, w0: Tensor, cached_x: Tensor, L: int):
    """NonlinAttention.streaming_forward (one head: the weights of head 0)."""
    T, B, _ = x.shape
    s, u, y = m.in_proj(x).chunk(3, dim=2)
    u = (u * torch.tanh(s)).transpose(0, 1).unsqueeze(0)                # (1,B,T,Ch)
    u = torch.cat([cached_x, u], dim=2)
    new_x = u[:, :, T:]
    o = torch.matmul(w0, u)[0].transpose(0, 1) * y                      # (T,B,Ch)
    return zk.linear(o, m.out_proj.weight, m.out_proj.bias, residual=x), new_x


def _conv_module(m, x: Tensor, cache: Tensor, kpm: Tensor):
    """ConvolutionModule.streaming_forward + ChunkCausalDepthwiseConv1d.streaming_forward."""
    T, B, _ = x.shape
    left = cache.shape[-1]
    g, s = m.in_proj(x).chunk(2, dim=2)
    g = g * torch.sigmoid(s)
    if kpm is not None:
        g = g.masked_fill(kpm.t().unsqueeze(-1), 0.0)
    hist = torch.cat([cache.permute(2, 0, 1), g], dim=0)                # (left+T,B,C)
    new_cache = hist[T:].permute(1, 2, 0)
    n = 1 + (left + T - 1) // T                                         # chunks of length T
    seq = F.pad(hist, (0, 0, 0, 0, n * T - (left + T), 0))
    y = zk.glu_chunk_causal_dwconv(seq, None, None, m.depthwise_conv, T)[(n - 1) * T:]
    return m.out_proj(y, residual=x), new_cache


def _layer(layer, src: Tensor, pos_proj_of, st: List[Tensor], L: int, kpm: Tensor):
    """Zipformer2EncoderLayer.streaming_forward."""
    ck, cna, cv1, cv2, cc1, cc2 = st
    orig = src
    w, ck = _attn_weights(layer.self_attn_weights, src, pos_proj_of(layer.self_attn_weights), ck,
                          L, kpm)
    cur = None if kpm is None else kpm[:, L:]
    src = layer.feed_forward1(src, residual=src)
    src, cna = _nonlin_attention(layer.nonlin_attention, src, w[0:1], cna, L)
    src, cv1 = _self_attn(layer.self_attn1, src, w, cv1, L)
    src, cc1 = _conv_module(layer.conv_module1, src, cc1, cur)
    src = layer.feed_forward2(src, residual=src)
    src = layer.bypass_mid(orig, src)
    src, cv2 = _self_attn(layer.self_attn2, src, w, cv2, L)
    src, cc2 = _conv_module(layer.conv_module2, src, cc2, cur)
    src = layer.feed_forward3(src, residual=src)
    src = layer.bypass(orig, layer.norm(src))
    return src, [ck, cna, cv1, cv2, cc1, cc2]


def _stack(enc, src: Tensor, states: List[Tensor], L: int, kpm: Tensor):
    """Zipformer2Encoder.streaming_forward."""
    T = src.shape[0]
    table = enc.encoder_pos.table(L + T, src.device)                    # offsets -(S-1) .. S-1

    def pos_proj_of(attn):
        return attn.linear_pos(table)

    new = []
    for i, layer in enumerate(enc.layers):
        src, st = _layer(layer, src, pos_proj_of, states[6 * i:6 * i + 6], L, kpm)
        new += st
    return src, new


def _embed(emb, x: Tensor, cached_left_pad: Tensor):
    """Conv2dSubsampling.streaming_forward + ConvNeXt.streaming_forward, channel-last.
    x (N,T,F) -> (N,(T-7)//2-3,D); cached_left_pad (N,C,3,F') as in the reference."""
    x = x.unsqueeze(-1)
    for m in emb.conv:
        if isinstance(m, torch.nn.Conv2d):
            x = zk.conv3x3_nhwc(x, m.weight, m.bias, m.stride, pad_w=m.padding[1])
            x = zk.swoosh(x, False)
    cn = emb.convnext
    T = x.shape[1] - cn.padding[0]
    bypass = x[:, :T]
    x = torch.cat([cached_left_pad.permute(0, 2, 3, 1), x], dim=1)
    new_pad = x[:, T:T + cn.padding[0]].permute(0, 3, 1, 2).contiguous()
    # "same"-padded depthwise conv; rows pad .. pad+T are the reference's valid-in-time outputs
    x = zk.dwconv2d_nhwc(x.contiguous(), cn.depthwise_conv.weight, cn.depthwise_conv.bias)
    x = x[:, cn.padding[0]:cn.padding[0] + T]
    x = zk.linear_big_m(x, cn.pointwise_conv1.weight.flatten(1), cn.pointwise_conv1.bias)
    x = zk.swoosh(x, True)
    x = zk.linear_big_m(x, cn.pointwise_conv2.weight.flatten(1), cn.pointwise_conv2.bias)
    x = bypass + x
    b, t, f, c = x.shape
    w = emb.out.weight.view(-1, c, f).permute(0, 2, 1).reshape(-1, f * c)
    x = zk.linear(x.reshape(b, t, f * c), w, emb.out.bias)
    return emb.out_norm(x), new_pad


# ------------------------------------------------------------------ entry points
@torch.no_grad()
def streaming_step(model, x: Tensor, states: List[Tensor]) -> Tuple[Tensor, List[Tensor]]:
    """x (N, 2*chunk+13, F) on the GPU; states from get_init_states / the previous call.
    -> (out (N, chunk//2, max(encoder_dim)), or log-softmax CTC scores when for_ctc; new states)."""
    if model.training:
        raise RuntimeError("streaming_step is an inference path: call model.eval() first")
    if not x.is_cuda:
        raise RuntimeError("streaming_step runs on the HIP kernels: x and states must be on cuda")
    chunk, left = model.chunk_size[0], model.left_context_frames[0]
    if chunk <= 0 or left < 0:
        raise ValueError("streaming needs a causal model with chunk_size > 0 and "
                         "left_context_frames >= 0")
    N = x.size(0)
    T = 2 * chunk + 13                              # 7 + 2*3: subsampling + ConvNeXt right context
    if x.size(1) != T:
        raise ValueError(f"streaming_step expects {T} frames per call, got {x.size(1)}")
    x, new_pad = _embed(model._encoder_embed, x.float(), states[-2])
    assert x.size(1) == chunk, (x.size(1), chunk)

    processed = states[-1]
    pm = (processed.unsqueeze(1) <= torch.arange(left, device=x.device).expand(N, left)).flip(1)
    kpm = torch.cat([pm, torch.zeros(N, chunk, dtype=torch.bool, device=x.device)], dim=1)
    new_processed = processed + chunk

    x = x.permute(1, 0, 2).contiguous()
    outputs, new_states, off = [], [], 0
    for i, stack in enumerate(model.encoders):
        ds, nl = model.downsampling_factor[i], stack.num_layers
        x = convert_num_channels(x, model.encoder_dim[i])
        st = states[6 * off:6 * (off + nl)]
        off += nl
        k_i = kpm[..., ::ds].contiguous()
        if ds == 1:
            x, st = _stack(stack, x, st, left // ds, k_i)
        else:
            orig = x
            y, st = _stack(stack.encoder, stack.downsample(x), st, left // ds, k_i)
            x = zk.bypass_upsampled(orig, y, stack.out_combiner._get_bypass_scale(y.shape[1]),
                                    stack.upsample.upsample)
        outputs.append(x)
        new_states += st
    x = model.downsample_output(model._get_full_dim_output(outputs)).permute(1, 0, 2)
    if model._for_ctc:
        x = F.log_softmax(model._ctc_projection(x), dim=-1)
    return x, new_states + [new_pad, new_processed]


@torch.no_grad()
def simulated_streaming_forward(model, x: Tensor, x_lens: Tensor, chunk_size=(32,),
                                left_context_frames=(128,)):
    """Zipformer2.streaming_forward (reference :391-406): whole-utterance forward with the chunked
    attention mask, 30 frames of log(1e-10) right padding; switches the model to causal chunks."""
    import math
    model.causal = True
    model.chunk_size = tuple(chunk_size)
    model.left_context_frames = tuple(left_context_frames)
    pad_len = 30
    x = F.pad(x, pad=(0, 0, 0, pad_len), value=math.log(1e-10))
    return model.forward(x, x_lens + pad_len)


class StreamingSession:
    """A stream of fixed-shape `streaming_step` calls replayed as ONE hipGraph.

    A chunk step is a few hundred small launches (every tensor is chunk-sized), so eager
    execution is bound by launch latency, not by the GPU.  All shapes are static from one call
    to the next, so the step is captured once -- input, states and output live in fixed buffers,
    the state update is part of the graph -- and each `step()` is a copy-in plus one graph launch.
    The first two steps of a session run eagerly on the capture stream (GEMM plan selection and
    position tables happen there); states then restart from `get_init_states`."""

    def __init__(self, model, batch_size: int = 1, device=None, warmup: int = 2):
        if model.training:
            raise RuntimeError("StreamingSession is an inference path: call model.eval() first")
        device = torch.device("cuda") if device is None else torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("StreamingSession runs on the HIP kernels: device must be cuda")
        self.model = model
        chunk = model.chunk_size[0]
        self.frames = 2 * chunk + 13
        self.x = torch.zeros(batch_size, self.frames, model._feature_dim, device=device)
        self.states = get_init_states(model, batch_size, device)
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            st = [s.clone() for s in self.states]
            for _ in range(warmup):
                _, st = streaming_step(model, self.x, st)
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side):
            self.out, new = streaming_step(model, self.x, self.states)
            for dst, src in zip(self.states, new):
                dst.copy_(src)

    def reset(self):
        for s in self.states:
            s.zero_()

    @torch.no_grad()
    def step(self, x: Tensor) -> Tensor:
        """x (B, 2*chunk+13, F) -> out (B, chunk//2, D) (a view of the session's output buffer:
        consume or clone it before the next step)."""
        if tuple(x.shape) != tuple(self.x.shape):
            raise ValueError(f"expected input of shape {tuple(self.x.shape)}, got {tuple(x.shape)}")
        self.x.copy_(x)
        self.graph.replay()
        return self.out
