"""Oracle: Zipformer2 encoder (+ gradient-shaping ops), functional torch-CPU fp32 restatement.

Restates model/encoder/zipformer.py (Zipformer2.forward :319-389 and the modules it calls),
model/layer/scaling.py (Balancer :719-902, Whiten :949-1095, LimitParamValue :1153-1190,
penalize_abs_values_gt :905-935, BiasNorm :347-476, ChunkCausalDepthwiseConv1d :552-681,
Swoosh :1340-1509, ActivationDropoutAndLinear :1512-1583) and model/layer/subsampling.py
(:26-178, :181-319) as plain functions over a state_dict that uses the reference's own
parameter names.  ScheduledFloat values are the `default`s, because the reference never
sets batch_count (scaling.py:182,192-196).

PINNED: tests/golden/zipformer_*.npz hold state_dict + inputs + outputs (+ gradients in the
deterministic training mode) captured from the reference classes by tools/gen_golden.py
(glog/onnx/k2.swoosh stand-ins only; swoosh formulas are the in-tree ones).

Randomness: `Ctl.rand` stands for Python's random.random() at each reference call site;
torch.rand / dropout draws use the global torch CPU generator in the reference's order, so
the same torch.manual_seed reproduces the reference's masks.
"""
import math

import torch
import torch.nn.functional as F


class Ctl:
    """training: bool; rand: callable replacing random.random(); chunk_size fixed by caller."""

    def __init__(self, training=False, rand=None, pos_dropout=0.15):
        self.training = training
        self.rand = rand if rand is not None else (lambda: 1.0)
        self.pos_dropout = pos_dropout


# ------------------------------------------------------------------ activations
def swoosh_l(x):
    return torch.logaddexp(torch.zeros((), dtype=x.dtype), x - 4.0) - 0.08 * x - 0.035


def swoosh_r(x):
    return torch.logaddexp(torch.zeros((), dtype=x.dtype), x - 1.0) - 0.08 * x - 0.313261687


# ------------------------------------------------------------------ gradient shaping
class _Balancer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, min_mean, max_mean, min_rms, max_rms, grad_scale, channel_dim):
        ctx.save_for_backward(x)
        ctx.cfg = (min_mean, max_mean, min_rms, max_rms, grad_scale, channel_dim % x.ndim)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        min_mean, max_mean, min_rms, max_rms, grad_scale, cd = ctx.cfg
        with torch.enable_grad():
            x = x.detach().float().requires_grad_(True)
            dims = [i for i in range(x.ndim) if i != cd]
            uvar = (x ** 2).mean(dim=dims, keepdim=True)
            mean = x.mean(dim=dims, keepdim=True)
            std = (uvar - mean * mean).clamp(min=1.0e-20).sqrt()
            rms = uvar.clamp(min=1.0e-20).sqrt()
            m = mean / std
            loss = (m - m.clamp(min=min_mean, max=max_mean)).abs() + \
                (rms.clamp(min=min_rms, max=max_rms) / rms).log().abs()
            loss.backward(gradient=torch.ones_like(loss))
        lg = x.grad
        lg_rms = (lg ** 2).mean(dim=dims, keepdim=True).sqrt().clamp(min=1.0e-20)
        lg = lg * (grad_scale / lg_rms)
        return g + g.abs() * lg, None, None, None, None, None, None


def _prop_pos_to_mean(x):
    x = -1 + 2 * x
    eps = 1.0e-10
    return 0.8139535143 * (math.log(1 + x + eps) - math.log(1 - x + eps)) / 2.0


def balancer(x, ctl, channel_dim, min_positive=0.05, max_positive=0.95, min_abs=0.2,
             max_abs=100.0, grad_scale=0.04, prob=0.4):
    if not x.requires_grad:
        return x
    if ctl.rand() < prob:
        return _Balancer.apply(x, _prop_pos_to_mean(min_positive), _prop_pos_to_mean(max_positive),
                               1.25331413732 * min_abs, 1.25331413732 * max_abs, grad_scale,
                               channel_dim)
    return x


def whitening_metric(x, num_groups):
    x = x.reshape(-1, x.shape[-1])
    n, c = x.shape
    cg = c // num_groups
    x = x.reshape(n, num_groups, cg).transpose(0, 1)
    x = x - x.mean(dim=1, keepdim=True)
    cov = torch.matmul(x.transpose(1, 2), x)
    mean_diag = torch.diagonal(cov, dim1=1, dim2=2).mean()
    covsq = (cov ** 2).sum() / (num_groups * cg)
    return covsq / (mean_diag ** 2 + 1.0e-20)


class _Whiten(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, num_groups, limit, grad_scale):
        ctx.save_for_backward(x)
        ctx.cfg = (num_groups, limit, grad_scale)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        num_groups, limit, grad_scale = ctx.cfg
        with torch.enable_grad():
            xd = x.detach().float().requires_grad_(True)
            metric = whitening_metric(xd, num_groups)
            if metric < limit:
                return g, None, None, None
            metric.backward()
        pg = xd.grad
        scale = grad_scale * (g.float().norm() / (pg.norm() + 1.0e-20))
        return g + pg * scale, None, None, None


def whiten(x, ctl, num_groups, limit, grad_scale, prob=0.25):
    if not x.requires_grad or ctl.rand() > prob or grad_scale == 0:
        return x
    return _Whiten.apply(x, num_groups, limit, grad_scale)


class _LimitParam(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lo, hi):
        ctx.save_for_backward(x)
        ctx.lo, ctx.hi = lo, hi
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = g * torch.where(torch.logical_and(g > 0, x < ctx.lo), -1.0, 1.0)
        g = g * torch.where(torch.logical_and(g < 0, x > ctx.hi), -1.0, 1.0)
        return g, None, None


def limit_param_value(x, ctl, lo, hi, prob=0.6, training=True):
    if training and ctl.rand() < prob:
        return _LimitParam.apply(x, lo, hi)
    return x


class _WithLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        ctx.ys = y.shape
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g, torch.ones(ctx.ys, dtype=g.dtype)


def penalize_abs_values_gt(x, limit, penalty):
    aux = penalty * ((x.sign() * ((x.abs() - limit) > 0)).to(torch.int8) * x)
    return _WithLoss.apply(x, aux)


class _ScaleGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, a):
        ctx.a = a
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.a, None


# ------------------------------------------------------------------ small modules
def bias_norm(x, bias, log_scale, ctl):
    ls = limit_param_value(log_scale, ctl, -1.5, 1.5, training=ctl.training)
    scales = (torch.mean((x - bias) ** 2, dim=-1, keepdim=True) ** -0.5) * ls.exp()
    return x * scales


def bypass(sd, pfx, src_orig, src, ctl):
    scale = sd[pfx + "bypass_scale"]
    if ctl.training:
        scale = limit_param_value(scale, ctl, 0.0, 1.0)
    return src_orig + (src - src_orig) * scale


def simple_downsample(src, bias, ds):
    T, B, C = src.shape
    dT = (T + ds - 1) // ds
    pad = dT * ds - T
    src = torch.cat((src, src[T - 1:].expand(pad, B, C)), dim=0)
    src = src.reshape(dT, ds, B, C)
    w = bias.softmax(dim=0).unsqueeze(-1).unsqueeze(-1)
    return (src * w).sum(dim=1)


def simple_upsample(src, up):
    T, B, C = src.shape
    return src.unsqueeze(1).expand(T, up, B, C).reshape(T * up, B, C)


def rel_pos_encoding(T, pos_dim):
    """CompactRelPositionalEncoding.extend_pe / forward (zipformer.py:1765-1833): (2T-1, pos_dim)."""
    x = torch.arange(-(T - 1), T).to(torch.float32).unsqueeze(1)
    freqs = 1 + torch.arange(pos_dim // 2)
    cl = pos_dim ** 0.5
    xc = cl * x.sign() * ((x.abs() + cl).log() - math.log(cl))
    ls = pos_dim / (2.0 * math.pi)
    xa = (xc / ls).atan()
    pe = torch.zeros(x.shape[0], pos_dim)
    pe[:, 0::2] = (xa * freqs).cos()
    pe[:, 1::2] = (xa * freqs).sin()
    pe[:, -1] = 1.0
    return pe


def chunk_causal_dwconv(sd, pfx, x, chunk_size, K):
    """x (B,C,T) -> (B,C,T)  (scaling.py:622-681)."""
    B, C, T = x.shape
    left = K // 2
    if chunk_size < 0 or chunk_size > T:
        chunk_size = T
    right = -T % chunk_size
    xp = F.pad(x, (left, right))
    x_causal = F.conv1d(xp[..., :left + T], sd[pfx + "causal_conv.weight"],
                        sd[pfx + "causal_conv.bias"], groups=C)
    xc = xp[..., left:]
    nch = xc.shape[2] // chunk_size
    xc = xc.reshape(B, C, nch, chunk_size).permute(0, 2, 1, 3).reshape(B * nch, C, chunk_size)
    xc = F.conv1d(xc, sd[pfx + "chunkwise_conv.weight"], sd[pfx + "chunkwise_conv.bias"],
                  padding=K // 2, groups=C)
    le, re = sd[pfx + "chunkwise_conv_scale"][0], sd[pfx + "chunkwise_conv_scale"][1]
    if chunk_size < K:
        le, re = le[:, :chunk_size], re[:, -chunk_size:]
    else:
        z = torch.zeros(C, chunk_size - K)
        le, re = torch.cat((le, z), -1), torch.cat((z, re), -1)
    xc = xc * (1.0 + (le + re))
    xc = xc.reshape(B, nch, C, chunk_size).permute(0, 2, 1, 3).reshape(B, C, nch * chunk_size)
    return xc[..., :T] + x_causal


# ------------------------------------------------------------------ layer pieces
def attn_weights(sd, pfx, x, pos_emb, H, qd, pd, attn_mask, kpm, ctl):
    T, B, _ = x.shape
    x = F.linear(x, sd[pfx + "in_proj.weight"], sd[pfx + "in_proj.bias"])
    q, k, p = x[..., :H * qd], x[..., H * qd:2 * H * qd], x[..., 2 * H * qd:]
    k = balancer(k, ctl, -1, 0.4, 0.6, 0.0, 100.0, prob=0.025)
    k = whiten(k, ctl, H, 3.0, 0.025)
    q = q.reshape(T, B, H, qd).permute(2, 1, 0, 3)
    p = p.reshape(T, B, H, pd).permute(2, 1, 0, 3)
    k = k.reshape(T, B, H, qd).permute(2, 1, 3, 0)
    scores = torch.matmul(q, k)
    if (not ctl.training) or ctl.rand() >= 0.0:
        pe = F.linear(pos_emb, sd[pfx + "linear_pos.weight"])          # (1, 2T-1, H*pd)
        pe = pe.reshape(-1, 2 * T - 1, H, pd).permute(2, 0, 3, 1)
        ps = torch.matmul(p, pe)                                       # (H,B,T,2T-1)
        idx = (T - 1) - torch.arange(T).unsqueeze(1) + torch.arange(T).unsqueeze(0)
        ps = ps[:, :, torch.arange(T).unsqueeze(1), idx]               # rel -> abs
        scores = scores + ps
    if ctl.training and ctl.rand() < 0.1:
        scores = penalize_abs_values_gt(scores, 25.0, 1.0e-04)
    if attn_mask is not None:
        scores = scores.masked_fill(attn_mask, -1000)
    if kpm is not None:
        scores = scores.masked_fill(kpm.unsqueeze(1), -1000)
    return scores.softmax(dim=-1)


def feed_forward(sd, pfx, x, ctl):
    x = F.linear(x, sd[pfx + "in_proj.weight"], sd[pfx + "in_proj.bias"])
    x = balancer(x, ctl, -1, 0.3, 1.0, 0.75, 5.0)
    x = F.linear(swoosh_l(x), sd[pfx + "out_proj.weight"], sd[pfx + "out_proj.bias"])
    return whiten(x, ctl, 1, 7.5, 0.01)


def self_attn(sd, pfx, x, w, ctl):
    T, B, _ = x.shape
    H = w.shape[0]
    x = F.linear(x, sd[pfx + "in_proj.weight"], sd[pfx + "in_proj.bias"])
    x = x.reshape(T, B, H, -1).permute(2, 1, 0, 3)
    x = torch.matmul(w, x).permute(2, 1, 0, 3).reshape(T, B, -1)
    x = F.linear(x, sd[pfx + "out_proj.weight"], sd[pfx + "out_proj.bias"])
    return whiten(x, ctl, 1, 7.5, 0.01)


def nonlin_attention(sd, pfx, x, w, ctl):
    T, B, _ = x.shape
    x = F.linear(x, sd[pfx + "in_proj.weight"], sd[pfx + "in_proj.bias"])
    s, x, y = x.chunk(3, dim=2)
    s = balancer(s, ctl, -1, 0.0, 0.0, 0.5, 5.0)
    s = torch.tanh(s)
    x = whiten(x, ctl, 1, 5.0, 0.01)
    x = x * s
    Hh = w.shape[0]
    x = x.reshape(T, B, Hh, -1).permute(2, 1, 0, 3)
    x = torch.matmul(w, x).permute(2, 1, 0, 3).reshape(T, B, -1)
    x = x * y
    x = F.linear(x, sd[pfx + "out_proj.weight"], sd[pfx + "out_proj.bias"])
    return whiten(x, ctl, 1, 5.0, 0.01)


def conv_module(sd, pfx, x, kpm, chunk_size, K, ctl):
    x = F.linear(x, sd[pfx + "in_proj.weight"], sd[pfx + "in_proj.bias"])
    x, s = x.chunk(2, dim=2)
    s = balancer(s, ctl, -1, 0.0, 1.0, 1.5, 1.0)
    x = x * torch.sigmoid(s)
    x = x.permute(1, 2, 0)
    if kpm is not None:
        x = x.masked_fill(kpm.unsqueeze(1).expand_as(x), 0.0)
    x = chunk_causal_dwconv(sd, pfx + "depthwise_conv.", x, chunk_size, K)
    x = balancer(x, ctl, 1, 0.0, 1.0, 0.0, 10.0)
    x = x.permute(2, 0, 1)
    x = whiten(x, ctl, 1, 7.5, 0.01)
    return F.linear(swoosh_r(x), sd[pfx + "out_proj.weight"], sd[pfx + "out_proj.bias"])


def encoder_layer(sd, pfx, src, pos_emb, cfg_i, chunk_size, attn_mask, kpm, ctl):
    H, qd, pd, K = cfg_i["H"], cfg_i["qd"], cfg_i["pd"], cfg_i["K"]
    orig = src
    w = attn_weights(sd, pfx + "self_attn_weights.", src, pos_emb, H, qd, pd, attn_mask, kpm, ctl)
    src = src + feed_forward(sd, pfx + "feed_forward1.", src, ctl)
    sel = w[0:1]
    if ctl.training:
        ctl.rand()          # const_attention_rate draw (rate 0: never taken)
    na = balancer(nonlin_attention(sd, pfx + "nonlin_attention.", src, sel, ctl), ctl, -1, 0.3, 0.7,
                  0.0, 100.0, prob=0.05)
    src = src + na
    src = src + self_attn(sd, pfx + "self_attn1.", src, w, ctl)
    src = src + conv_module(sd, pfx + "conv_module1.", src, kpm, chunk_size, K, ctl)
    src = src + balancer(feed_forward(sd, pfx + "feed_forward2.", src, ctl), ctl, -1, 0.3, 0.7,
                         0.0, 2.0, prob=0.05)
    src = bypass(sd, pfx + "bypass_mid.", orig, src, ctl)
    src = src + self_attn(sd, pfx + "self_attn2.", src, w, ctl)
    src = src + conv_module(sd, pfx + "conv_module2.", src, kpm, chunk_size, K, ctl)
    src = src + balancer(feed_forward(sd, pfx + "feed_forward3.", src, ctl), ctl, -1, 0.3, 0.7,
                         0.0, 4.0, prob=0.05)
    src = balancer(src, ctl, -1, 0.45, 0.55, 0.2, 4.0)
    src = bias_norm(src, sd[pfx + "norm.bias"], sd[pfx + "norm.log_scale"], ctl)
    src = bypass(sd, pfx + "bypass.", orig, src, ctl)
    src = balancer(src, ctl, -1, 0.45, 0.55, 0.1, 4.0)
    return whiten(src, ctl, 1, 4.0, 0.01)


def encoder_stack(sd, pfx, src, n_layers, cfg_i, pos_dim, chunk_size, fmask, attn_mask, kpm, ctl):
    T = src.shape[0]
    pos_emb = rel_pos_encoding(T, pos_dim).unsqueeze(0)
    pos_emb = F.dropout(pos_emb, p=ctl.pos_dropout, training=ctl.training)
    out = src * fmask
    for l in range(n_layers):
        out = encoder_layer(sd, f"{pfx}layers.{l}.", out, pos_emb, cfg_i, chunk_size, attn_mask,
                            kpm, ctl)
        out = out * fmask
    return out


def conv2d_subsampling(sd, pfx, x, x_lens, ctl):
    x = x.unsqueeze(1)
    x = F.conv2d(x, sd[pfx + "conv.0.weight"], sd[pfx + "conv.0.bias"], padding=(0, 1))
    if ctl.training:
        x = _ScaleGrad.apply(x, 0.2)
    x = balancer(x, ctl, 1, max_abs=1.0)
    x = swoosh_r(x)
    x = F.conv2d(x, sd[pfx + "conv.4.weight"], sd[pfx + "conv.4.bias"], stride=2)
    x = balancer(x, ctl, 1, max_abs=4.0)
    x = swoosh_r(x)
    x = F.conv2d(x, sd[pfx + "conv.7.weight"], sd[pfx + "conv.7.bias"], stride=(1, 2))
    x = balancer(x, ctl, 1, max_abs=4.0)
    x = swoosh_r(x)
    # ConvNeXt
    byp = x
    c = x.shape[1]
    x = F.conv2d(x, sd[pfx + "convnext.depthwise_conv.weight"],
                 sd[pfx + "convnext.depthwise_conv.bias"], padding=(3, 3), groups=c)
    x = F.conv2d(x, sd[pfx + "convnext.pointwise_conv1.weight"],
                 sd[pfx + "convnext.pointwise_conv1.bias"])
    x = balancer(x, ctl, 1, 0.3, 1.0, 0.75, 5.0)
    x = swoosh_l(x)
    x = F.conv2d(x, sd[pfx + "convnext.pointwise_conv2.weight"],
                 sd[pfx + "convnext.pointwise_conv2.bias"])
    x = byp + x
    x = balancer(x, ctl, 1, 0.4, 0.6, 1.0, 6.0)
    if x.requires_grad:
        x = whiten(x.transpose(1, 3), ctl, 1, 5.0, 0.01).transpose(1, 3)
    b, c, t, f = x.shape
    x = x.transpose(1, 2).reshape(b, t, c * f)
    x = F.linear(x, sd[pfx + "out.weight"], sd[pfx + "out.bias"])
    x = whiten(x, ctl, 1, 4.0, 0.02)
    x = bias_norm(x, sd[pfx + "out_norm.bias"], sd[pfx + "out_norm.log_scale"], ctl)
    return x, (x_lens - 7) // 2


def zipformer_forward(sd, cfg, x, x_lens, ctl, chunk_size=-1, left_context_chunks=-1):
    """cfg: dict(downsampling_factor, num_encoder_layers, encoder_dim, encoder_unmasked_dim,
    num_heads, query_head_dim, pos_head_dim, cnn_module_kernel, pos_dim) (tuples per stack).
    x (B,T,80), x_lens (B,) -> (B,T',max_dim), lengths."""
    ds_f, dims = cfg["downsampling_factor"], cfg["encoder_dim"]
    ns = len(ds_f)
    x, lens = conv2d_subsampling(sd, "_encoder_embed.", x, x_lens, ctl)
    kpm = torch.arange(int(lens.max())).unsqueeze(0) >= lens.unsqueeze(1)
    x = x.transpose(0, 1)
    T, B, _ = x.shape
    if ctl.training:
        m1 = (torch.rand(1, B, 1) > 0.125).to(x.dtype)
        m2 = torch.logical_and(m1, (torch.rand(1, B, 1) > 0.125).to(x.dtype))
        m = torch.cat((m1, m2), dim=-1)
        fmasks = []
        for i in range(ns):
            fm = torch.ones(1, B, dims[i])
            u1 = cfg["encoder_unmasked_dim"][i]
            u2 = u1 + (dims[i] - u1) // 2
            fm[:, :, u1:u2] *= m[..., 0:1]
            fm[:, :, u2:] *= m[..., 1:2]
            fmasks.append(fm)
    else:
        fmasks = [1.0] * ns
    attn_mask = None
    if chunk_size > 0:
        lcc = left_context_chunks if left_context_chunks >= 0 else 1000000
        c = torch.arange(T, dtype=torch.int32) // chunk_size
        attn_mask = torch.logical_or(c.unsqueeze(0) > c.unsqueeze(1),
                                     c.unsqueeze(0) < c.unsqueeze(1) - lcc)
    outs = []
    for i in range(ns):
        d = dims[i]
        x = x[..., :d] if d <= x.shape[-1] else torch.cat(
            (x, torch.zeros(*x.shape[:-1], d - x.shape[-1])), dim=-1)
        cfg_i = dict(H=cfg["num_heads"][i], qd=cfg["query_head_dim"][i], pd=cfg["pos_head_dim"][i],
                     K=cfg["cnn_module_kernel"][i])
        ds = ds_f[i]
        k_i = kpm[..., ::ds]
        if ds == 1:
            x = encoder_stack(sd, f"encoders.{i}.", x, cfg["num_encoder_layers"][i], cfg_i,
                              cfg["pos_dim"], chunk_size, fmasks[i], attn_mask, k_i, ctl)
        else:
            orig = x
            y = simple_downsample(x, sd[f"encoders.{i}.downsample.bias"], ds)
            am = attn_mask[::ds, ::ds] if attn_mask is not None else None
            y = encoder_stack(sd, f"encoders.{i}.encoder.", y, cfg["num_encoder_layers"][i], cfg_i,
                              cfg["pos_dim"], chunk_size // ds if chunk_size > 0 else -1,
                              fmasks[i], am, k_i, ctl)
            y = simple_upsample(y, ds)[:orig.shape[0]]
            x = bypass(sd, f"encoders.{i}.out_combiner.", orig, y, ctl)
        outs.append(x)
    pieces = [outs[-1]]
    cur = dims[-1]
    for i in range(ns - 2, -1, -1):
        if dims[i] > cur:
            pieces.append(outs[i][..., cur:dims[i]])
            cur = dims[i]
    x = torch.cat(pieces, dim=-1)
    x = simple_downsample(x, sd["downsample_output.bias"], 2)
    return x.transpose(0, 1), (lens + 1) // 2


# ------------------------------------------------------------------ streaming (inference only)
def streaming_init_states(cfg, B, left):
    """Zipformer2.get_init_states (zipformer.py:529-600): per layer (cached_key (L,B,H*qd),
    cached_nonlin_attn (1,B,L,3D/4), cached_val1, cached_val2 (L,B,H*vd), cached_conv1,
    cached_conv2 (B,D,K//2)), then the ConvNeXt left pad (B,128,3,F') and processed_lens (B,)."""
    states = []
    for i, ds in enumerate(cfg["downsampling_factor"]):
        D, H = cfg["encoder_dim"][i], cfg["num_heads"][i]
        L = left // ds
        for _ in range(cfg["num_encoder_layers"][i]):
            states += [torch.zeros(L, B, H * cfg["query_head_dim"][i]),
                       torch.zeros(1, B, L, 3 * D // 4),
                       torch.zeros(L, B, H * cfg["value_head_dim"][i]),
                       torch.zeros(L, B, H * cfg["value_head_dim"][i]),
                       torch.zeros(B, D, cfg["cnn_module_kernel"][i] // 2),
                       torch.zeros(B, D, cfg["cnn_module_kernel"][i] // 2)]
    fw = (((cfg.get("feature_dim", 80) - 1) // 2) - 1) // 2
    states.append(torch.zeros(B, 128, 3, fw))
    states.append(torch.zeros(B, dtype=torch.int64))
    return states


def _stream_attn_weights(sd, pfx, x, pos_emb, cached_key, L, H, qd, pd, kpm):
    """RelPositionMultiheadAttentionWeights.streaming_forward (zipformer.py:2079-2190)."""
    T, B, _ = x.shape
    x = F.linear(x, sd[pfx + "in_proj.weight"], sd[pfx + "in_proj.bias"])
    q, k, p = x[..., :H * qd], x[..., H * qd:2 * H * qd], x[..., 2 * H * qd:]
    k = torch.cat([cached_key, k], dim=0)
    cached_key = k[-L:]
    S = k.shape[0]
    q = q.reshape(T, B, H, qd).permute(2, 1, 0, 3)
    p = p.reshape(T, B, H, pd).permute(2, 1, 0, 3)
    k = k.reshape(S, B, H, qd).permute(2, 1, 3, 0)
    scores = torch.matmul(q, k)
    pe = F.linear(pos_emb, sd[pfx + "linear_pos.weight"])              # (1, L+2T-1, H*pd)
    pe = pe.reshape(-1, 2 * T - 1 + L, H, pd).permute(2, 0, 3, 1)
    ps = torch.matmul(p, pe)                                           # (H,B,T,L+2T-1)
    idx = (T - 1) - torch.arange(T).unsqueeze(1) + torch.arange(S).unsqueeze(0)
    ps = ps[:, :, torch.arange(T).unsqueeze(1), idx]                   # rel -> abs, (H,B,T,S)
    scores = scores + ps
    scores = scores.masked_fill(kpm.unsqueeze(1), -1000)
    return scores.softmax(dim=-1), cached_key


def _stream_self_attn(sd, pfx, x, w, cached_val, L):
    """SelfAttention.streaming_forward (zipformer.py:2282-2333)."""
    T, B, _ = x.shape
    H = w.shape[0]
    x = F.linear(x, sd[pfx + "in_proj.weight"], sd[pfx + "in_proj.bias"])
    x = torch.cat([cached_val, x], dim=0)
    cached_val = x[-L:]
    x = x.reshape(T + L, B, H, -1).permute(2, 1, 0, 3)
    x = torch.matmul(w, x).permute(2, 1, 0, 3).reshape(T, B, -1)
    return F.linear(x, sd[pfx + "out_proj.weight"], sd[pfx + "out_proj.bias"]), cached_val


def _stream_nonlin_attention(sd, pfx, x, w, cached_x, L):
    """NonlinAttention.streaming_forward (zipformer.py:2485-2541)."""
    T, B, _ = x.shape
    x = F.linear(x, sd[pfx + "in_proj.weight"], sd[pfx + "in_proj.bias"])
    s, x, y = x.chunk(3, dim=2)
    x = x * torch.tanh(s)
    Hh = w.shape[0]
    x = x.reshape(T, B, Hh, -1).permute(2, 1, 0, 3)
    xp = torch.cat([cached_x, x], dim=2)
    cached_x = xp[:, :, -L:]
    x = torch.matmul(w, xp).permute(2, 1, 0, 3).reshape(T, B, -1)
    x = x * y
    return F.linear(x, sd[pfx + "out_proj.weight"], sd[pfx + "out_proj.bias"]), cached_x


def _stream_conv_module(sd, pfx, x, cache, kpm, K):
    """ConvolutionModule.streaming_forward (zipformer.py:2697-2741) with
    ChunkCausalDepthwiseConv1d.streaming_forward (scaling.py:683-716)."""
    x = F.linear(x, sd[pfx + "in_proj.weight"], sd[pfx + "in_proj.bias"])
    x, s = x.chunk(2, dim=2)
    x = (x * torch.sigmoid(s)).permute(1, 2, 0)                        # (B,C,T)
    x = x.masked_fill(kpm.unsqueeze(1).expand_as(x), 0.0)
    B, C, T = x.shape
    d = pfx + "depthwise_conv."
    left = K // 2
    x = torch.cat([cache, x], dim=2)
    cache = x[..., -left:]
    x_causal = F.conv1d(x, sd[d + "causal_conv.weight"], sd[d + "causal_conv.bias"], groups=C)
    xc = F.conv1d(x[..., left:], sd[d + "chunkwise_conv.weight"], sd[d + "chunkwise_conv.bias"],
                  padding=K // 2, groups=C)
    le, re = sd[d + "chunkwise_conv_scale"][0], sd[d + "chunkwise_conv_scale"][1]
    if T < K:
        le, re = le[:, :T], re[:, -T:]
    else:
        z = torch.zeros(C, T - K)
        le, re = torch.cat((le, z), -1), torch.cat((z, re), -1)
    x = (xc * (1.0 + (le + re)) + x_causal).permute(2, 0, 1)
    return F.linear(swoosh_r(x), sd[pfx + "out_proj.weight"], sd[pfx + "out_proj.bias"]), cache


def _stream_layer(sd, pfx, src, pos_emb, st, L, cfg_i, kpm, ctl):
    """Zipformer2EncoderLayer.streaming_forward (zipformer.py:1223-1338)."""
    H, qd, pd, K = cfg_i["H"], cfg_i["qd"], cfg_i["pd"], cfg_i["K"]
    ck, cna, cv1, cv2, cc1, cc2 = st
    orig = src
    w, ck = _stream_attn_weights(sd, pfx + "self_attn_weights.", src, pos_emb, ck, L, H, qd, pd, kpm)
    src = src + feed_forward(sd, pfx + "feed_forward1.", src, ctl)
    na, cna = _stream_nonlin_attention(sd, pfx + "nonlin_attention.", src, w[0:1], cna, L)
    src = src + na
    sa, cv1 = _stream_self_attn(sd, pfx + "self_attn1.", src, w, cv1, L)
    src = src + sa
    cv, cc1 = _stream_conv_module(sd, pfx + "conv_module1.", src, cc1, kpm[:, L:], K)
    src = src + cv
    src = src + feed_forward(sd, pfx + "feed_forward2.", src, ctl)
    src = bypass(sd, pfx + "bypass_mid.", orig, src, ctl)
    sa, cv2 = _stream_self_attn(sd, pfx + "self_attn2.", src, w, cv2, L)
    src = src + sa
    cv, cc2 = _stream_conv_module(sd, pfx + "conv_module2.", src, cc2, kpm[:, L:], K)
    src = src + cv
    src = src + feed_forward(sd, pfx + "feed_forward3.", src, ctl)
    src = bias_norm(src, sd[pfx + "norm.bias"], sd[pfx + "norm.log_scale"], ctl)
    src = bypass(sd, pfx + "bypass.", orig, src, ctl)
    return src, [ck, cna, cv1, cv2, cc1, cc2]


def _stream_stack(sd, pfx, src, states, n_layers, cfg_i, pos_dim, L, kpm, ctl):
    """Zipformer2Encoder.streaming_forward (zipformer.py:1432-1496); pos_emb covers offsets
    -(T+L-1) .. T-1 (CompactRelPositionalEncoding.forward :1815-1833)."""
    T = src.shape[0]
    pe = rel_pos_encoding(T + L, pos_dim)
    c = pe.shape[0] // 2
    pos_emb = pe[c - (T + L) + 1:c + T].unsqueeze(0)
    new = []
    for l in range(n_layers):
        src, st = _stream_layer(sd, f"{pfx}layers.{l}.", src, pos_emb, states[6 * l:6 * l + 6], L,
                                cfg_i, kpm, ctl)
        new += st
    return src, new


def _stream_subsampling(sd, pfx, x, cached_left_pad, ctl):
    """Conv2dSubsampling.streaming_forward (subsampling.py:321-375) with ConvNeXt.streaming_forward
    (:134-178): T input frames -> (T-7)//2 - 3 output frames."""
    x = x.unsqueeze(1)
    x = swoosh_r(F.conv2d(x, sd[pfx + "conv.0.weight"], sd[pfx + "conv.0.bias"], padding=(0, 1)))
    x = swoosh_r(F.conv2d(x, sd[pfx + "conv.4.weight"], sd[pfx + "conv.4.bias"], stride=2))
    x = swoosh_r(F.conv2d(x, sd[pfx + "conv.7.weight"], sd[pfx + "conv.7.bias"], stride=(1, 2)))
    T = x.shape[2] - 3
    byp = x[:, :, :T]
    x = torch.cat([cached_left_pad, x], dim=2)
    cached_left_pad = x[:, :, T:T + 3]
    c = x.shape[1]
    x = F.conv2d(x, sd[pfx + "convnext.depthwise_conv.weight"],
                 sd[pfx + "convnext.depthwise_conv.bias"], padding=(0, 3), groups=c)
    x = F.conv2d(x, sd[pfx + "convnext.pointwise_conv1.weight"],
                 sd[pfx + "convnext.pointwise_conv1.bias"])
    x = F.conv2d(swoosh_l(x), sd[pfx + "convnext.pointwise_conv2.weight"],
                 sd[pfx + "convnext.pointwise_conv2.bias"])
    x = byp + x
    b, c, t, f = x.shape
    x = x.transpose(1, 2).reshape(b, t, c * f)
    x = F.linear(x, sd[pfx + "out.weight"], sd[pfx + "out.bias"])
    x = bias_norm(x, sd[pfx + "out_norm.bias"], sd[pfx + "out_norm.log_scale"], ctl)
    return x, cached_left_pad


def streaming_step(sd, cfg, x, states, chunk, left, for_ctc=False):
    """Zipformer2.streaming_step (zipformer.py:601-663) + _encoder_layer_streaming_forward
    (:465-527).  x (B, 2*chunk+13, F); states as streaming_init_states.  Returns
    (out (B, chunk//2, max_dim) or log-softmax CTC scores, new_states)."""
    ctl = Ctl(training=False)
    ds_f, dims = cfg["downsampling_factor"], cfg["encoder_dim"]
    B = x.shape[0]
    assert x.shape[1] == 2 * chunk + 13
    x, new_pad = _stream_subsampling(sd, "_encoder_embed.", x, states[-2], ctl)
    assert x.shape[1] == chunk
    processed = states[-1]
    pm = (processed.unsqueeze(1) <= torch.arange(left).expand(B, left)).flip(1)
    kpm = torch.cat([pm, torch.zeros(B, chunk, dtype=torch.bool)], dim=1)
    new_processed = processed + chunk
    x = x.permute(1, 0, 2)
    outs, new_states, off = [], [], 0
    for i, ds in enumerate(ds_f):
        d, nl = dims[i], cfg["num_encoder_layers"][i]
        x = x[..., :d] if d <= x.shape[-1] else torch.cat(
            (x, torch.zeros(*x.shape[:-1], d - x.shape[-1])), dim=-1)
        cfg_i = dict(H=cfg["num_heads"][i], qd=cfg["query_head_dim"][i], pd=cfg["pos_head_dim"][i],
                     K=cfg["cnn_module_kernel"][i])
        st = states[6 * off:6 * (off + nl)]
        off += nl
        k_i = kpm[..., ::ds]
        if ds == 1:
            x, st = _stream_stack(sd, f"encoders.{i}.", x, st, nl, cfg_i, cfg["pos_dim"], left // ds,
                                  k_i, ctl)
        else:
            orig = x
            y = simple_downsample(x, sd[f"encoders.{i}.downsample.bias"], ds)
            y, st = _stream_stack(sd, f"encoders.{i}.encoder.", y, st, nl, cfg_i, cfg["pos_dim"],
                                  left // ds, k_i, ctl)
            y = simple_upsample(y, ds)[:orig.shape[0]]
            x = bypass(sd, f"encoders.{i}.out_combiner.", orig, y, ctl)
        outs.append(x)
        new_states += st
    pieces = [outs[-1]]
    cur = dims[-1]
    for i in range(len(ds_f) - 2, -1, -1):
        if dims[i] > cur:
            pieces.append(outs[i][..., cur:dims[i]])
            cur = dims[i]
    x = torch.cat(pieces, dim=-1)
    x = simple_downsample(x, sd["downsample_output.bias"], 2).permute(1, 0, 2)
    if for_ctc:
        x = F.log_softmax(F.linear(x, sd["_ctc_projection.weight"], sd["_ctc_projection.bias"]),
                          dim=-1)
    return x, new_states + [new_pad, new_processed]
