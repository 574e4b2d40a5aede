"""Conformer-block entry points (the seam between model/encoder/conformer.py and the HIP ABI).

Raw helpers (`ln_fwd`, `ln_bwd`, `silu_fwd`, ...) launch one kernel each on plain (rows, C)
tensors and are what the layer executor (conf_layer.py) schedules by hand; the
`torch.autograd.Function`s wrap the same launches for the module-by-module path (evaluation,
dropout > 0, parameters outside a FlatStore).  Reference: torchaudio.models.Conformer as called
at model/encoder/conformer.py:170-178,193 (block structure restated in csrc/conf_elem.hip).
There is no CPU path: CPU tensors raise.
"""
import ctypes
import math

import torch

from . import _native as N
from . import flat

_F32 = torch.float32


def _rows(x):
    C = x.shape[-1]
    x2 = x.reshape(-1, C)
    if x2.dtype != _F32 or not x2.is_contiguous() or x2.data_ptr() % 16:
        x2 = x2.contiguous().float()
        if x2.data_ptr() % 16:
            x2 = x2.clone()
    return x2


# ------------------------------------------------------------------ raw launches
def ln_fwd(x2, y2, alpha, weight, bias, eps):
    """-> (xsum | None, out, stats).  y2 given: the input is x2 + alpha * y2 (also returned)."""
    if not x2.is_cuda:
        raise RuntimeError("speech2text_amd conformer kernels run on the GPU only (no CPU fallback)")
    R, C = x2.shape
    out = torch.empty_like(x2)
    stats = torch.empty((R, 2), dtype=_F32, device=x2.device)
    xsum = torch.empty_like(x2) if y2 is not None else None
    N.PROF[0] and N.profile_note("s2t_layernorm_fwd", 4.0 * R * C * (2 if y2 is None else 4))
    N.check(N.lib().s2t_layernorm_fwd(N.fp(x2), N.fp(y2), float(alpha), N.fp(weight), N.fp(bias),
                                      R, C, float(eps), N.fp(xsum), N.fp(out), N.fp(stats),
                                      N.stream()), "s2t_layernorm_fwd")
    return xsum, out, stats


class LnFold(ctypes.Structure):
    """Mirror of S2tLnFold (include/s2t_mi355.h)."""
    _fields_ = [("partial", ctypes.c_void_p), ("rows", ctypes.c_long),
                ("dgamma", ctypes.c_void_p), ("dbeta", ctypes.c_void_p)]


def ln_bwd(x2, stats, weight, dy2, resid2, pend):
    """-> dx (+ resid2).  The per-workgroup sums of d gamma / d beta are parked in a scratch
    buffer and (partial, rows) is appended to `pend`; ln_param_grad(pend items) folds them."""
    R, C = x2.shape
    dx = torch.empty_like(x2)
    partial = torch.empty(N.lib().s2t_layernorm_bwd_partial_floats(R, C), dtype=_F32,
                          device=x2.device)
    N.PROF[0] and N.profile_note("s2t_layernorm_bwd", 4.0 * R * C * (3 if resid2 is None else 4))
    N.check(N.lib().s2t_layernorm_bwd(N.fp(x2), N.fp(stats), N.fp(weight), N.fp(dy2),
                                      N.fp(resid2), R, C, N.fp(dx), N.fp(partial), N.stream()),
            "s2t_layernorm_bwd")
    pend.append((partial, R))
    return dx


def ln_param_grad(items, C):
    """items: [(partial, rows, dgamma (C,), dbeta (C,))] -- dgamma / dbeta += the folded partial
    sums of up to any number of LayerNorms of width C, one launch per 8."""
    arr = (LnFold * len(items))()
    for q, (partial, rows, dg, db) in zip(arr, items):
        q.partial, q.rows = partial.data_ptr(), rows
        q.dgamma, q.dbeta = dg.data_ptr(), db.data_ptr()
    N.PROF[0] and N.profile_note("s2t_layernorm_param_grad", 4.0 * sum(it[0].numel() for it in items) + 16.0 * C * len(items))
    N.check(N.lib().s2t_layernorm_param_grad(len(items), ctypes.cast(arr, ctypes.c_void_p), C,
                                             N.stream()), "s2t_layernorm_param_grad")


def silu_fwd(h2, p=0.0, seed=0):
    """silu(h), or drop(silu(h)) with the hashed mask of (seed, p) (nn.Dropout after the SiLU)."""
    a = torch.empty_like(h2)
    N.PROF[0] and N.profile_note("s2t_silu_drop_fwd" if p > 0.0 else "s2t_silu_fwd", 8.0 * h2.numel())
    if p > 0.0:
        N.check(N.lib().s2t_silu_drop_fwd(N.fp(h2), h2.numel(), float(p), int(seed), N.fp(a),
                                          N.stream()), "s2t_silu_drop_fwd")
    else:
        N.check(N.lib().s2t_silu_fwd(N.fp(h2), h2.numel(), N.fp(a), N.stream()), "s2t_silu_fwd")
    return a


def silu_bwd(h2, da2, scale=1.0, inplace=True, p=0.0, seed=0):
    """scale * da * silu'(h) (* the dropout mask of (seed, p)); written over da2 when `inplace`."""
    dh = da2 if inplace else torch.empty_like(da2)
    N.PROF[0] and N.profile_note("s2t_silu_drop_bwd" if p > 0.0 else "s2t_silu_bwd", 12.0 * h2.numel())
    if p > 0.0:
        N.check(N.lib().s2t_silu_drop_bwd(N.fp(h2), N.fp(da2), h2.numel(), float(scale), float(p),
                                          int(seed), N.fp(dh), N.stream()), "s2t_silu_drop_bwd")
    else:
        N.check(N.lib().s2t_silu_bwd(N.fp(h2), N.fp(da2), h2.numel(), float(scale), N.fp(dh),
                                     N.stream()), "s2t_silu_bwd")
    return dh


def dropout_add(x2, y2, alpha, p, seed):
    """x + alpha * drop(y) in one pass (x2 None: alpha * drop(y), the gradient through the site);
    mask = the stateless hash of (seed, element index), kept values scaled by 1 / (1 - p)."""
    out = torch.empty_like(y2)
    N.PROF[0] and N.profile_note("s2t_dropout_add", 4.0 * y2.numel() * (3 if x2 is not None else 2))
    N.check(N.lib().s2t_dropout_add(N.fp(x2), N.fp(y2), y2.numel(), float(alpha), float(p),
                                    int(seed), N.fp(out), N.stream()), "s2t_dropout_add")
    return out


_BN_WS = {}


def _bn_ws(dev, C):
    ws = _BN_WS.get((dev, C))
    if ws is None:
        ws = _BN_WS[(dev, C)] = torch.empty(N.lib().s2t_bn_workspace_floats(C), dtype=_F32,
                                            device=dev)
    return ws


def bn_silu_fwd(x2, bn):
    """training-mode nn.BatchNorm1d + SiLU on (rows, C) -> (y, save_mean, save_rstd); updates the
    module's running statistics like torch does."""
    R, C = x2.shape
    y = torch.empty_like(x2)
    mean = torch.empty(C, dtype=_F32, device=x2.device)
    rstd = torch.empty(C, dtype=_F32, device=x2.device)
    track = bn.track_running_stats and bn.running_mean is not None
    if bn.momentum is None:          # torch: cumulative moving average, factor 1 / (batches seen incl. this one)
        mom = 1.0 / (float(bn.num_batches_tracked) + 1.0) if track else 0.0
    else:
        mom = float(bn.momentum)
    N.PROF[0] and N.profile_note("s2t_bn_silu_fwd", 12.0 * R * C)
    N.check(N.lib().s2t_bn_silu_fwd(N.fp(x2), N.fp(bn.weight), N.fp(bn.bias), float(bn.eps), mom,
                                    N.fp(bn.running_mean) if track else None,
                                    N.fp(bn.running_var) if track else None,
                                    N.lp(bn.num_batches_tracked) if track else None, R, C,
                                    N.fp(y), N.fp(mean), N.fp(rstd), N.fp(_bn_ws(x2.device, C)),
                                    N.stream()), "s2t_bn_silu_fwd")
    return y, mean, rstd


def bn_silu_eval(x2, bn):
    R, C = x2.shape
    y = torch.empty_like(x2)
    rstd = torch.rsqrt(bn.running_var + bn.eps)
    N.check(N.lib().s2t_bn_silu_apply(N.fp(x2), N.fp(bn.running_mean), N.fp(rstd), N.fp(bn.weight),
                                      N.fp(bn.bias), R, C, N.fp(y), N.stream()),
            "s2t_bn_silu_apply")
    return y


def bn_silu_bwd(x2, ds2, mean, rstd, weight, bias, dgamma, dbeta):
    R, C = x2.shape
    dx = torch.empty_like(x2)
    N.PROF[0] and N.profile_note("s2t_bn_silu_bwd", 20.0 * R * C)
    N.check(N.lib().s2t_bn_silu_bwd(N.fp(x2), N.fp(ds2), N.fp(mean), N.fp(rstd), N.fp(weight),
                                    N.fp(bias), R, C, N.fp(dx), N.fp(dgamma), N.fp(dbeta),
                                    N.fp(_bn_ws(x2.device, C)), N.stream()), "s2t_bn_silu_bwd")
    return dx


def mhsa_fwd(qkv2, lens, T, B, H, dropout_p=0.0, seed=0):
    """qkv2 (T*B, 3D) rows (t, b) -> (o (T*B, D), lse (B,H,T))."""
    D = qkv2.shape[1] // 3
    dh = D // H
    o = torch.empty((T * B, D), dtype=_F32, device=qkv2.device)
    lse = torch.empty((B, H, T), dtype=_F32, device=qkv2.device)
    N.PROF[0] and N.profile_note("s2t_mhsa_fwd", 4.0 * (qkv2.numel() + o.numel()), 4.0 * B * H * T * T * dh)
    N.check(N.lib().s2t_mhsa_fwd(N.fp(qkv2), 3 * D, 0, D, 2 * D, N.lp(lens), T, B, H, dh,
                                 1.0 / math.sqrt(dh), float(dropout_p), int(seed), N.fp(o), D,
                                 N.fp(lse), N.stream()), "s2t_mhsa_fwd")
    return o, lse


def mhsa_bwd(qkv2, lens, T, B, H, o2, do2, lse, dropout_p=0.0, seed=0):
    D = qkv2.shape[1] // 3
    dh = D // H
    dqkv = torch.empty_like(qkv2)
    delta = torch.empty((B, H, T), dtype=_F32, device=qkv2.device)
    N.PROF[0] and N.profile_note("s2t_mhsa_bwd", 4.0 * (2 * qkv2.numel() + 2 * o2.numel()),
                   14.0 * B * H * T * T * dh)
    N.check(N.lib().s2t_mhsa_bwd(N.fp(qkv2), 3 * D, 0, D, 2 * D, N.lp(lens), T, B, H, dh,
                                 1.0 / math.sqrt(dh), float(dropout_p), int(seed), N.fp(o2),
                                 N.fp(do2), D, N.fp(lse), N.fp(delta), N.fp(dqkv), N.stream()),
            "s2t_mhsa_bwd")
    return dqkv


def _grad_slots(params):
    """Flat-store gradient views of `params` if all of them live in one, else None."""
    out = []
    for p in params:
        if not (p.is_leaf and flat.owned(p) and p.grad is not None and p.grad.is_contiguous()):
            return None
        out.append(p.grad)
    return out


# ------------------------------------------------------------------ autograd wrappers
class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        x2 = _rows(x)
        _, out, stats = ln_fwd(x2, None, 0.0, weight.contiguous(), bias.contiguous(), eps)
        ctx.save_for_backward(x2, stats, weight)
        ctx.params = (weight, bias)
        return out.view(x.shape)

    @staticmethod
    def backward(ctx, g):
        x2, stats, weight = ctx.saved_tensors
        g2 = _rows(g)
        slots = _grad_slots(ctx.params)
        pend = []
        dx = ln_bwd(x2, stats, weight.contiguous(), g2, None, pend)
        C = x2.shape[1]
        if slots is not None:
            ln_param_grad([pend[0] + (slots[0], slots[1])], C)
            return dx.view(g.shape), None, None, None
        acc = torch.zeros((2, C), dtype=_F32, device=x2.device)
        ln_param_grad([pend[0] + (acc[0], acc[1])], C)
        return dx.view(g.shape), acc[0], acc[1], None


def layer_norm(x, ln):
    if not x.is_cuda:
        raise RuntimeError("speech2text_amd.layer_norm needs device tensors (HIP path only)")
    C = x.shape[-1]
    if C % 4 or C > 1024 or x.dtype != _F32 or ln.weight is None:
        # outside the kernel's rules (row length a multiple of 4, <= 1024, affine): torch's device
        # kernel -- still the GPU, never a host path
        return torch.nn.functional.layer_norm(x, (C,), ln.weight, ln.bias, ln.eps)
    return _LayerNorm.apply(x, ln.weight, ln.bias, ln.eps)


class _SiLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h):
        h2 = _rows(h)
        ctx.save_for_backward(h2)
        return silu_fwd(h2).view(h.shape)

    @staticmethod
    def backward(ctx, g):
        (h2,) = ctx.saved_tensors
        return silu_bwd(h2, _rows(g), 1.0, inplace=False).view(g.shape)


def silu(x):
    if not x.is_cuda:
        raise RuntimeError("speech2text_amd.silu needs device tensors (HIP path only)")
    if x.numel() % 4 or x.dtype != _F32:
        return torch.nn.functional.silu(x)             # (device kernel of torch: odd sizes / dtypes)
    return _SiLU.apply(x)


class _BnSilu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, bn):
        x2 = _rows(x)
        y, mean, rstd = bn_silu_fwd(x2, bn)
        ctx.save_for_backward(x2, mean, rstd, weight, bias)
        ctx.params = (weight, bias)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, g):
        x2, mean, rstd, weight, bias = ctx.saved_tensors
        g2 = _rows(g)
        slots = _grad_slots(ctx.params)
        if slots is not None:
            dx = bn_silu_bwd(x2, g2, mean, rstd, weight, bias, slots[0], slots[1])
            return dx.view(g.shape), None, None, None
        acc = torch.zeros((2, x2.shape[1]), dtype=_F32, device=x2.device)
        dx = bn_silu_bwd(x2, g2, mean, rstd, weight, bias, acc[0], acc[1])
        return dx.view(g.shape), acc[0], acc[1], None


def batchnorm_silu(x, bn):
    """SiLU(BatchNorm1d(x)) on channel-last (..., C): batch statistics over every leading index."""
    if not x.is_cuda:
        raise RuntimeError("speech2text_amd.batchnorm_silu needs device tensors (HIP path only)")
    if bn.training or not bn.track_running_stats:
        return _BnSilu.apply(x, bn.weight, bn.bias, bn)
    return bn_silu_eval(_rows(x), bn).view(x.shape)


class _Mhsa(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, lens, H, p, seed):
        T, B, D3 = qkv.shape
        q2 = _rows(qkv)
        o, lse = mhsa_fwd(q2, lens, T, B, H, p, seed)
        ctx.save_for_backward(q2, o, lse, lens)
        ctx.cfg = (T, B, H, p, seed)
        return o.view(T, B, D3 // 3)

    @staticmethod
    def backward(ctx, g):
        q2, o, lse, lens = ctx.saved_tensors
        T, B, H, p, seed = ctx.cfg
        dqkv = mhsa_bwd(q2, lens, T, B, H, o, _rows(g), lse, p, seed)
        return dqkv.view(T, B, -1), None, None, None, None


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        ctx.cfg = (p, seed)
        x2 = x.contiguous()
        return dropout_add(None, x2.view(-1), 1.0, p, seed).view(x.shape)

    @staticmethod
    def backward(ctx, g):
        p, seed = ctx.cfg
        g2 = g.contiguous().float()
        return dropout_add(None, g2.view(-1), 1.0, p, seed).view(g.shape), None, None


class Dropout(torch.nn.Dropout):
    """nn.Dropout whose training pass on the GPU is the hashed-mask kernel (one pass each way, no
    mask tensor, no device RNG state); same `p`, same module type for isinstance checks."""

    def forward(self, x):
        p = float(self.p)
        if (not self.training or p == 0.0 or not x.is_cuda or x.dtype != _F32 or x.numel() % 4
                or p >= 1.0):
            return super().forward(x)
        return _Dropout.apply(x, p, draw_seed())


def draw_seed():
    """Host-side 62-bit seed for the in-kernel dropout hash (torch's CPU generator: follows
    torch.manual_seed, never touches the device)."""
    return int(torch.randint(0, 2 ** 62, (1,)).item())


def mhsa(qkv, lengths, num_heads, dropout_p=0.0):
    """qkv (T,B,3D) = in_proj(x) -> (T,B,D): softmax(q k^T / sqrt(dh), keys >= lengths[b] masked) v
    per head; dropout_p: dropout on the attention probabilities (training)."""
    if not qkv.is_cuda:
        raise RuntimeError("speech2text_amd.mhsa needs device tensors (HIP path only)")
    lens = None if lengths is None else lengths.to(device=qkv.device, dtype=torch.int64).contiguous()
    T, B, D3 = qkv.shape
    dh = D3 // 3 // num_heads
    if dh not in (16, 32, 64) or qkv.dtype != _F32:
        # head widths the flash kernel is not built for: torch's device attention, same masking
        q, k, v = (t.reshape(T, B, num_heads, dh).permute(1, 2, 0, 3) for t in qkv.chunk(3, dim=-1))
        mask = None
        if lens is not None:
            mask = (torch.arange(T, device=qkv.device)[None, :] < lens[:, None])[:, None, None, :]
        o = torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=dropout_p)
        return o.permute(2, 0, 1, 3).reshape(T, B, D3 // 3)
    seed = draw_seed() if dropout_p > 0.0 else 0
    return _Mhsa.apply(qkv, lens, num_heads, float(dropout_p), seed)


# ------------------------------------------------------------------ layer-norm LSTM layer
class _LnLstm(torch.autograd.Function):
    """hs, hT, cT = LSTM(gx; p2g, g_norm, c_norm) for a whole sequence (csrc/lstm.hip): gx (T,B,4H)
    = x2g(x).  Backward: one reverse-time kernel gives the gradient w.r.t. the raw gates; the
    recurrent weight's gradient is a TN GEMM of it against the shifted hidden states."""

    @staticmethod
    def forward(ctx, gx, wp, gg, gb, cg, cb, eps, h0, c0):
        T, B, G = gx.shape
        H = G // 4
        dev = gx.device
        gx = gx.contiguous().float()
        wp_t = wp.detach().t().contiguous()
        hs = torch.empty((T, B, H), dtype=_F32, device=dev)
        ghat = torch.empty((T, B, G), dtype=_F32, device=dev)
        chat = torch.empty((T, B, H), dtype=_F32, device=dev)
        rstd = torch.empty((T, B, 2), dtype=_F32, device=dev)
        hT = torch.empty((B, H), dtype=_F32, device=dev)
        cT = torch.empty((B, H), dtype=_F32, device=dev)
        h0c = None if h0 is None else h0.contiguous().float()
        c0c = None if c0 is None else c0.contiguous().float()
        if T == 0:                   # nothing to scan: the final state is the initial one
            hT = torch.zeros_like(hT) if h0c is None else h0c.clone()
            cT = torch.zeros_like(cT) if c0c is None else c0c.clone()
        N.check(N.lib().s2t_lnlstm_fwd(N.fp(gx), N.fp(wp_t), N.fp(gg), N.fp(gb), N.fp(cg), N.fp(cb),
                                       N.fp(h0c), N.fp(c0c), T, B, H, float(eps), N.fp(hs),
                                       N.fp(ghat), N.fp(chat), N.fp(rstd), N.fp(hT), N.fp(cT),
                                       N.stream()), "s2t_lnlstm_fwd")
        ctx.save_for_backward(wp, gg, gb, cg, cb, ghat, chat, rstd, hs, h0c, c0c)
        ctx.params = (wp, gg, gb, cg, cb)
        ctx.mark_non_differentiable(hT, cT)
        return hs, hT, cT

    @staticmethod
    def backward(ctx, dhs, _dh, _dc):
        from . import zip_kernels as zk
        wp, gg, gb, cg, cb, ghat, chat, rstd, hs, h0c, c0c = ctx.saved_tensors
        wparam, ggp, gbp, cgp, cbp = ctx.params
        T, B, G = ghat.shape
        H = G // 4
        dev = ghat.device
        dhs = dhs.contiguous().float()
        dgx = torch.empty_like(ghat)
        ln = gg is not None
        slots = _grad_slots([ggp, gbp, cgp, cbp]) if ln else None
        if ln and slots is None:
            acc = torch.zeros(2 * G + 2 * H, dtype=_F32, device=dev)
            slots = [acc[:G], acc[G:2 * G], acc[2 * G:2 * G + H], acc[2 * G + H:]]
            ret_ln = tuple(slots)
        else:
            ret_ln = (None, None, None, None)
        wc = wp.contiguous()
        N.check(N.lib().s2t_lnlstm_bwd(N.fp(wc), N.fp(gg), N.fp(gb), N.fp(cg), N.fp(cb),
                                       N.fp(c0c), T, B, H, N.fp(ghat), N.fp(chat), N.fp(rstd),
                                       N.fp(dhs), N.fp(dgx), N.fp(slots[0]) if ln else None,
                                       N.fp(slots[1]) if ln else None,
                                       N.fp(slots[2]) if ln else None,
                                       N.fp(slots[3]) if ln else None, N.stream()),
                "s2t_lnlstm_bwd")
        # d p2g.weight = sum_t dg_t^T h_{t-1}
        hprev = torch.empty_like(hs)
        hprev[1:] = hs[:-1]
        if h0c is None:
            hprev[0].zero_()
        else:
            hprev[0] = h0c
        g2, a2 = dgx.view(T * B, G), hprev.view(T * B, H)
        dwp = None
        if not zk.wgrad_into(wparam, None, g2, a2):
            dwp, _ = zk.linear_wgrad(g2, a2, False)
        return (dgx, dwp) + ret_ln + (None, None, None)


def lnlstm(gx, p2g_weight, g_norm, c_norm, h0=None, c0=None):
    """gx (T,B,4H) -> (hs (T,B,H), h_T, c_T).  g_norm / c_norm: nn.LayerNorm modules, or
    nn.Identity (no layer norm)."""
    if not gx.is_cuda:
        raise RuntimeError("speech2text_amd.lnlstm needs device tensors (HIP path only)")
    if isinstance(g_norm, torch.nn.LayerNorm):
        return _LnLstm.apply(gx, p2g_weight, g_norm.weight, g_norm.bias, c_norm.weight,
                             c_norm.bias, g_norm.eps, h0, c0)
    return _LnLstm.apply(gx, p2g_weight, None, None, None, None, 0.0, h0, c0)


# ------------------------------------------------------------------ Subsampling: first conv + ReLU
def conv1_relu_ok(conv, x):
    C = conv.out_channels
    return (x.is_cuda and conv.in_channels == 1 and conv.kernel_size == (3, 3)
            and conv.stride == (2, 2) and conv.padding == (0, 0) and conv.bias is not None
            and C % 4 == 0 and C <= 1024 and 256 % (C // 4) == 0 and x.shape[-1] >= 3
            and x.shape[-2] >= 3)


class _Conv1Relu(torch.autograd.Function):
    """relu(Conv2d(1, C, 3, stride 2)(x)) for x (B,T,F) -> (B,C,T1,F1) in channels_last memory
    (csrc/conf_front.hip).  Backward: weight / bias gradients from one pass over the incoming
    gradient (ReLU mask recomputed); the input gradient (features never need one in training) is
    formed with torch ops when a test asks for it."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x3 = x.contiguous().float()
        B, T, F_ = x3.shape
        C = weight.shape[0]
        T1, F1 = (T - 3) // 2 + 1, (F_ - 3) // 2 + 1
        out = torch.empty((B, T1, F1, C), dtype=_F32, device=x.device)
        w2 = weight.detach().reshape(C, 9).contiguous()
        b1 = bias.detach().contiguous()
        N.PROF[0] and N.profile_note("s2t_conv1_relu_fwd", 4.0 * (x3.numel() + out.numel()))
        N.check(N.lib().s2t_conv1_relu_fwd(N.fp(x3), N.fp(w2), N.fp(b1), B, T, F_, C, N.fp(out),
                                           N.stream()), "s2t_conv1_relu_fwd")
        ctx.save_for_backward(x3, w2, b1)
        ctx.params = (weight, bias)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        x3, w2, b1 = ctx.saved_tensors
        weight, bias = ctx.params
        B, T, F_ = x3.shape
        C = w2.shape[0]
        d = g.permute(0, 2, 3, 1)
        if not d.is_contiguous() or d.dtype != _F32:
            d = d.contiguous().float()
        slots = _grad_slots([weight, bias])
        if slots is None:
            acc = torch.zeros(10 * C, dtype=_F32, device=x3.device)
            dw, db = acc[:9 * C], acc[9 * C:]
        else:
            dw, db = slots
        ws = torch.empty(N.lib().s2t_conv1_relu_workspace_floats(C), dtype=_F32, device=x3.device)
        N.PROF[0] and N.profile_note("s2t_conv1_relu_wgrad", 4.0 * (x3.numel() + d.numel()))
        N.check(N.lib().s2t_conv1_relu_wgrad(N.fp(x3), N.fp(w2), N.fp(b1), N.fp(d), B, T, F_, C,
                                             N.raw(dw), N.raw(db), N.fp(ws), N.stream()),
                "s2t_conv1_relu_wgrad")
        dx = None
        if ctx.needs_input_grad[0]:
            z = torch.nn.functional.conv2d(x3.unsqueeze(1), w2.view(C, 1, 3, 3), b1, stride=2)
            dx = torch.nn.grad.conv2d_input((B, 1, T, F_), w2.view(C, 1, 3, 3), g * (z > 0),
                                            stride=2).view(B, T, F_)
        if slots is None:
            return dx, dw.view(weight.shape), db
        return dx, None, None


def conv1_relu(x, conv):
    """x (B,T,F) -> relu(conv(x.unsqueeze(1))) as (B,C,T1,F1) channels_last."""
    return _Conv1Relu.apply(x, conv.weight, conv.bias)
