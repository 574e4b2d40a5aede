"""One autograd node per Zipformer2 encoder layer (training hot path).

The module-by-module form of the layer (model/encoder/zipformer.py, mirroring reference
model/encoder/zipformer.py:785-1010) costs autograd ~45 nodes and the host ~5 ms of Python per
layer and step -- more than the GPU needs for the layer's kernels.  Here the same launches are
issued from one forward and one hand-scheduled backward:

  * the layer's random gradient-shaping decisions (Balancer / Whiten / limit_param_value /
    positional skip / score penalty) are drawn FIRST, in exactly the order the module path draws
    them, from the same generator (scaling._rand);
  * forward keeps the residual stream x0..x11 as (T*B, D) row blocks, every module's residual add
    rides in its last GEMM's epilogue, Whiten statistics are computed where the module fires;
  * backward walks the modules in reverse: data gradients on the current stream (the residual
    branch rides in each module's first-projection dgrad epilogue), weight/bias gradients as TN
    GEMMs accumulated straight into the flat gradient buffer on the side stream, the small
    per-channel parameters committed by one kernel each, the attention-weights gradient
    contracted from its three consumers without materialising (H,B,T,T).

Cases the executor does not cover (masks, per-utterance skip masks, dropout, the 10 % score
penalty draw, parameters outside a FlatStore) run the module path; the values already drawn are
queued for it so the random stream is consumed identically.
"""
import ctypes
import os

import torch

from . import _native as N
from . import flat
from . import zip_kernels as zk
from . import zip_native as zn
from .model.layer import scaling as S

_F32 = torch.float32
ENABLED = os.environ.get("S2T_LAYER_EXEC", "1") == "1"
CALLS = [0]          # layer calls served by the executor (tests assert the path really ran)
STATS = {"penalize": 0, "penalty_active": 0}     # score-penalty draws / calls where it was non-zero
PEN_LIMIT, PEN_VALUE = 25.0, 1.0e-4              # zipformer.py:2010-2026


class _Plan:
    """Decisions of one layer call, in draw order."""
    __slots__ = ("k_bal", "k_wh", "use_pos", "penalize", "ff1", "na", "sa1", "cv1", "ff2", "mid_lim", "sa2",
                 "cv2", "ff3", "bal1", "norm_lim", "byp_lim", "bal2", "wh_out")


def _bal(m):
    return S._rand() < float(m.prob)


def _wh(m):
    r = S._rand()
    return not (r > m.prob or float(m.grad_scale) == 0)


def _static_ok(layer):
    """Shape / structure conditions, evaluated once per layer object."""
    ok = layer.__dict__.get("_zl_static")
    if ok is None:
        sa = layer.self_attn_weights
        H = sa.num_heads
        D = layer.embed_dim
        dv1 = layer.self_attn1.in_proj.weight.shape[0] // H
        dv2 = layer.self_attn2.in_proj.weight.shape[0] // H
        dims = [D, sa.in_proj.weight.shape[0], H * sa.query_head_dim, H * sa.pos_head_dim,
                layer.nonlin_attention.hidden_channels, layer.self_attn1.in_proj.weight.shape[0],
                layer.self_attn2.in_proj.weight.shape[0], sa.linear_pos.weight.shape[1]]
        dims += [m.in_proj.weight.shape[0] for m in (layer.feed_forward1, layer.feed_forward2,
                                                     layer.feed_forward3)]
        ok = (all(d % 4 == 0 for d in dims) and dv1 <= 16 and dv2 <= 16 and dv1 + dv2 <= 32
              and sa.dropout == 0.0 and layer.norm.channel_dim in (-1, 2))
        layer.__dict__["_zl_static"] = ok
        if ok and _PIN_ATTRS:
            _pin_attrs(layer)
    return ok


_PIN_ATTRS = os.environ.get("S2T_PIN_ATTRS", "1") == "1"


def _pin_attrs(root):
    """Parameters and submodules of `root`'s tree also become plain instance attributes:
    nn.Module finds them through its Python-level __getattr__ (three dict probes, ~0.7 us), and a
    step of 12 layers asks 3 800 times.  Safe under this framework's standing invariant that
    parameter OBJECTS never change after setup (FlatStore and the fused optimizers rebind
    `.data` only); Module.__setattr__ drops the shortcut if a Parameter / Module name is
    re-assigned.  Buffers are NOT pinned: re-assigning one (or Module._apply replacing it) does not
    go through that path and would leave a stale shadow."""
    for m in root.modules():
        d = m.__dict__
        for table in (m._parameters, m._modules):
            for name, v in table.items():
                if v is not None and name not in d:
                    d[name] = v


def eligible(layer, src, attn_mask, key_padding_mask):
    if not (ENABLED and layer.training and torch.is_grad_enabled() and src.is_cuda
            and src.requires_grad and src.dtype == _F32 and src.dim() == 3
            and src.shape[0] >= 4):
        return False
    if not _static_ok(layer):
        return False
    for r in (layer.attention_skip_rate, layer.conv_skip_rate, layer.ff2_skip_rate,
              layer.ff3_skip_rate, layer.const_attention_rate, layer.bypass.skip_rate,
              layer.bypass.straight_through_rate, layer.bypass_mid.skip_rate,
              layer.bypass_mid.straight_through_rate, layer.feed_forward1.out_proj.dropout_p,
              layer.feed_forward2.out_proj.dropout_p, layer.feed_forward3.out_proj.dropout_p):
        if float(r) != 0.0:
            return False
    # every parameter must live in a FlatStore: the backward accumulates into its gradient views
    p = layer.bypass.bypass_scale
    return flat.owned(p) and p.grad is not None


_M8 = {}


def _mask8(m):
    """uint8 copy of a boolean mask; the same mask object serves every layer of a stack, so the
    last conversion of each is kept."""
    if m is None:
        return None
    hit = _M8.get(id(m))
    if hit is not None and hit[0] is m:
        return hit[1]
    if len(_M8) > 8:
        _M8.clear()
    m8 = m.to(torch.uint8).contiguous()
    _M8[id(m)] = (m, m8)
    return m8


def _pen_ok(sa, T):
    """The score-limit flag comes from the MFMA attention kernel only (zip_attn.hip)."""
    H, qd, pd = sa.num_heads, sa.query_head_dim, sa.pos_head_dim
    return (T <= 512 and pd <= 4 and qd % 8 == 0 and (H * (2 * qd + pd)) % 4 == 0
            and "S2T_ATTN_FWD_OLD" not in os.environ)


def run(layer, src, pos_emb, chunk_size, attn_mask=None, key_padding_mask=None, feature_mask=None):
    """-> layer output (times the stack's feature mask (1,B,D) if one is given), or None when this
    call's draws ask for the score penalty on a shape the flagging kernel does not serve."""
    sa = layer.self_attn_weights
    d = _Plan()
    r0, r1, r2, r3 = S._rand(), S._rand(), S._rand(), S._rand()
    d.penalize = r3 < 0.1                            # penalize_abs_values_gt on the raw scores
    if d.penalize and not _pen_ok(sa, src.shape[0]):
        S._REPLAY.extend((r0, r1, r2, r3))
        return None
    d.k_bal = r0 < float(sa.balance_keys.prob)
    d.k_wh = not (r1 > sa.whiten_keys.prob or float(sa.whiten_keys.grad_scale) == 0)
    d.use_pos = r2 >= float(sa.pos_emb_skip_rate)

    def ff(m, post):
        fb, fw = _bal(m.hidden_balancer), _wh(m.out_whiten)
        return fb, fw, (_bal(post) if post is not None else False)

    def cv(m):
        return _bal(m.balancer1), _bal(m.balancer2), _wh(m.whiten)

    d.ff1 = ff(layer.feed_forward1, None)
    S._rand()                                        # const_attention draw (rate is 0 here)
    na = layer.nonlin_attention
    d.na = (_bal(na.balancer), _wh(na.whiten1), _wh(na.whiten2), _bal(layer.balancer_na))
    d.sa1 = _wh(layer.self_attn1.whiten)
    d.cv1 = cv(layer.conv_module1)
    d.ff2 = ff(layer.feed_forward2, layer.balancer_ff2)
    d.mid_lim = S._rand() < 0.6
    d.sa2 = _wh(layer.self_attn2.whiten)
    d.cv2 = cv(layer.conv_module2)
    d.ff3 = ff(layer.feed_forward3, layer.balancer_ff3)
    d.bal1 = _bal(layer.balancer1)
    d.norm_lim = S._rand() < 0.6
    d.byp_lim = S._rand() < 0.6
    d.bal2 = _bal(layer.balancer2)
    d.wh_out = _wh(layer.whiten)
    CALLS[0] += 1
    fm = None
    if feature_mask is not None:
        fm = feature_mask.reshape(src.shape[1], src.shape[2])
        if fm.dtype != _F32 or not fm.is_contiguous():
            fm = fm.contiguous().float()
    a8, k8 = _mask8(attn_mask), _mask8(key_padding_mask)
    # one C call per pass (csrc/zip_layer.hip) once the layer's GEMM shapes have been timed;
    # the Python executor below is the same launch sequence and times them on first sight
    L = zn.usable(layer, src.shape[0], src.shape[1])
    if L is not None:
        return zn.run(layer, L, src, pos_emb, chunk_size, d, a8, k8, fm)
    return _LayerFn.apply(src, pos_emb, layer, chunk_size, d, a8, k8, fm)


# ----------------------------------------------------------------------------- raw helpers
def _e(rows, cols, dev):
    return torch.empty((rows, cols), dtype=_F32, device=dev)


_PEND = []


def _wgrad(w, b, g2, a2):
    """Queue dW += g2^T a2, db += colsum(g2): the layer's weight gradients go out as ONE grouped
    launch at the end of its backward (zk.wgrad_group), on the side stream."""
    _PEND.append((w, b, g2, a2))


def _whiten_bwd(mod, x, g, stats):
    out, active = zk.whiten_backward(x, g, stats, float(mod.whitening_limit),
                                     float(mod.grad_scale))
    mod.prob = mod.max_prob if active else mod.min_prob
    return out


def _balancer_bwd(mod, x, g, inplace=False, swoosh_l=None):
    return zk.balancer_backward(x, g, *mod.cfg(2), inplace=inplace, swoosh_l=swoosh_l)


class _Commit(ctypes.Structure):
    """Mirror of S2tCommit (include/s2t_mi355.h)."""
    _fields_ = [("x", ctypes.c_void_p), ("d", ctypes.c_void_p), ("grad", ctypes.c_void_p),
                ("lo", ctypes.c_float), ("hi", ctypes.c_float), ("limit", ctypes.c_int),
                ("n", ctypes.c_long)]


def _commit(items):
    """items: [(param, d, lo, hi, limit)]: param.grad += limit_param(d), d cleared -- one launch
    for all of them, then tell the gradient reducer."""
    arr = (_Commit * len(items))()
    for q, (p, d, lo, hi, limit) in zip(arr, items):
        q.x, q.d, q.grad = p.data_ptr(), d.data_ptr(), p.grad.data_ptr()
        q.lo, q.hi, q.limit, q.n = float(lo), float(hi), int(limit), d.numel()
    N.PROF[0] and N.profile_note("s2t_param_grad_commit_n", 12.0 * sum(it[1].numel() for it in items))
    N.check(N.lib().s2t_param_grad_commit_n(len(items), ctypes.cast(arr, ctypes.c_void_p),
                                            N.stream()), "s2t_param_grad_commit_n")
    for it in items:
        flat.grad_written(it[0])


_ACC = {}


def _layer_acc(dev, D):
    """Persistent accumulator of a layer's per-channel parameter gradients, [bypass scale |
    bypass_mid scale | norm bias | norm log_scale]; zeroed once, the commit kernel clears it."""
    acc = _ACC.get((dev, D))
    if acc is None:
        acc = _ACC[(dev, D)] = torch.zeros(3 * D + 4, dtype=_F32, device=dev)
    return acc


class _Saved:
    pass


# ----------------------------------------------------------------------------- modules
def _ff_fwd(m, dec, x_in):
    fb, fw, fp = dec
    sv = _Saved()
    # the activation is KEPT for the weight gradient (the reference recomputes it to save memory,
    # scaling.py:1512-1583; 288 GB of HBM make the ~1 GB per step the cheaper side of that trade);
    # it leaves the in-projection's epilogue as a second output (no separate Swoosh pass)
    sv.h, a = zk.lt_matmul(0, x_in, m.in_proj.weight, m.in_proj.bias, act2="swoosh_l")
    sv.a = a
    sv.y = sv.st = None
    if not (fw or fp):
        out = zk.lt_matmul(0, a, m.out_proj.weight, m.out_proj.bias, x_in)
    else:
        # the module's own output (for the Whiten / Balancer on it) AND the residual stream after
        # it from one launch
        sv.y, out = zk.lt_matmul(0, a, m.out_proj.weight, m.out_proj.bias, act2="add", resid_b=x_in)
        if fw:
            sv.st = zk.WhitenStats(sv.y, m.out_whiten.num_groups)
    return out, sv


def _ff_bwd(m, post, dec, sv, x_in, g):
    fb, fw, fp = dec
    gy = g
    if fp:
        gy = _balancer_bwd(post, sv.y, gy)
    if fw:
        gy = _whiten_bwd(m.out_whiten, sv.y, gy, sv.st)
    W = m.out_proj.weight
    _wgrad(W, m.out_proj.bias, gy, sv.a)
    if fb:                                   # Swoosh backward AND the Balancer's update in the dgrad epilogue
        dh = zk.lt_matmul(1, gy, W, act_src=sv.h, act_kind="swoosh_l", bal=m.hidden_balancer.cfg(2))
    else:                                    # Swoosh backward in the data-gradient GEMM's epilogue
        dh = zk.lt_matmul(1, gy, W, act_src=sv.h, act_kind="swoosh_l")
    _wgrad(m.in_proj.weight, m.in_proj.bias, dh, x_in)
    return zk.lt_matmul(1, dh, m.in_proj.weight, None, g)


def _sa_fwd(m, fw, x_in, W, T, B, H):
    sv = _Saved()
    sv.v = zk.lt_matmul(0, x_in, m.in_proj.weight, m.in_proj.bias)
    dv = sv.v.shape[1] // H
    sv.o = torch.empty_like(sv.v)
    N.PROF[0] and N.profile_note("s2t_attn_apply", 4.0 * (W.numel() + 2 * sv.v.numel()))
    N.check(N.lib().s2t_attn_apply(N.fp(W), N.fp(sv.v), T, B, H, dv, 0, N.fp(sv.o), N.stream()),
            "s2t_attn_apply")
    sv.y = sv.st = None
    if not fw:
        out = zk.lt_matmul(0, sv.o, m.out_proj.weight, m.out_proj.bias, x_in)
    else:
        sv.y, out = zk.lt_matmul(0, sv.o, m.out_proj.weight, m.out_proj.bias, act2="add", resid_b=x_in)
        sv.st = zk.WhitenStats(sv.y, m.whiten.num_groups)
    return out, sv


def _sa_bwd(m, fw, sv, x_in, g, W, T, B, H, pairs):
    gy = _whiten_bwd(m.whiten, sv.y, g, sv.st) if fw else g
    _wgrad(m.out_proj.weight, m.out_proj.bias, gy, sv.o)
    dO = zk.lt_matmul(1, gy, m.out_proj.weight)
    dv = sv.v.shape[1] // H
    dV = torch.empty_like(sv.v)
    N.PROF[0] and N.profile_note("s2t_attn_apply", 4.0 * (W.numel() + 2 * sv.v.numel()))
    N.check(N.lib().s2t_attn_apply(N.fp(W), N.fp(dO), T, B, H, dv, 1, N.fp(dV), N.stream()),
            "s2t_attn_apply(T)")
    pairs.append((dO, sv.v, sv.o, dv))
    _wgrad(m.in_proj.weight, m.in_proj.bias, dV, x_in)
    return zk.lt_matmul(1, dV, m.in_proj.weight, None, g)


def _conv_fwd(m, dec, x_in, T, B, chunk_size, k8):
    fb1, fb2, fw = dec
    sv = _Saved()
    D = x_in.shape[1]
    if chunk_size >= 0:
        assert m.causal, "Must initialize model with causal=True if you use chunk_size"
    sv.u = zk.lt_matmul(0, x_in, m.in_proj.weight, m.in_proj.bias)                          # (R, 2D)
    sv.cp = zk.conv_params(m.depthwise_conv, T, chunk_size)
    y, a = zk.zipconv_forward(sv.u.view(T, B, 2 * D), D, k8, *sv.cp, act=False)   # (y, SwooshR(y))
    sv.y, sv.a = y.view(T * B, D), a.view(T * B, D)
    sv.st = zk.WhitenStats(sv.y, m.whiten.num_groups) if fw else None
    return zk.lt_matmul(0, sv.a, m.out_proj.weight, m.out_proj.bias, x_in), sv


def _conv_bwd(m, dec, sv, x_in, g, T, B, k8):
    fb1, fb2, fw = dec
    D = x_in.shape[1]
    W = m.out_proj.weight
    _wgrad(W, m.out_proj.bias, g, sv.a)
    if fb2 and not fw:                       # Swoosh backward AND the Balancer's update in the dgrad epilogue
        dy = zk.lt_matmul(1, g, W, act_src=sv.y, act_kind="swoosh_r", bal=m.balancer2.cfg(2))
    else:                                    # Swoosh backward in the data-gradient GEMM's epilogue
        dy = zk.lt_matmul(1, g, W, act_src=sv.y, act_kind="swoosh_r")
        if fw:
            dy = _whiten_bwd(m.whiten, sv.y, dy, sv.st)
        if fb2:
            dy = _balancer_bwd(m.balancer2, sv.y, dy)
    chunk, K, wc, bc, wk, bk, scale = sv.cp
    plist = (wc, bc, wk, bk, scale)
    grads = zk.direct_grads(plist)
    if grads is None:
        raise RuntimeError("zip_layer: conv-module parameters must live in a FlatStore")
    du = zk.zipconv_backward(sv.u.view(T, B, 2 * D), D, k8, chunk, K, wc, wk, bk, scale,
                             dy.view(T, B, D), grads).view(T * B, 2 * D)
    for p in plist:
        if p is not None:
            flat.grad_written(p)
    if fb1:
        _balancer_bwd(m.balancer1, sv.u[:, D:], du[:, D:], inplace=True)
    _wgrad(m.in_proj.weight, m.in_proj.bias, du, x_in)
    return zk.lt_matmul(1, du, m.in_proj.weight, None, g)


def _na_fwd(m, dec, x_in, W, T, B):
    fb, fw1, fw2, fp = dec
    sv = _Saved()
    L, st = N.lib(), N.stream()
    dev = x_in.device
    sv.u = zk.lt_matmul(0, x_in, m.in_proj.weight, m.in_proj.bias)                          # (R, 3C) = [s|x|y]
    C = sv.u.shape[1] // 3
    sv.xs = torch.empty((B, T, C), dtype=_F32, device=dev)
    N.PROF[0] and N.profile_note("s2t_nonlin_gate_fwd", 12.0 * T * B * C)
    N.check(L.s2t_nonlin_gate_fwd(N.fp(sv.u), T, B, C, N.fp(sv.xs), st), "nonlin_gate_fwd")
    sv.wm = W[0]                                                             # (B,T,T)
    with zk.gemm_class(zk.CLS_F):
        sv.z = zk.batched_matmul(1, sv.wm, sv.xs)                            # W0 @ x, (B,T,C)
    sv.o = _e(T * B, C, dev)
    N.PROF[0] and N.profile_note("s2t_nonlin_out_fwd", 12.0 * T * B * C)
    N.check(L.s2t_nonlin_out_fwd(N.fp(sv.z), N.fp(sv.u), T, B, C, N.fp(sv.o), st),
            "nonlin_out_fwd")
    sv.st1 = zk.WhitenStats(sv.u[:, C:2 * C], m.whiten1.num_groups) if fw1 else None
    sv.y = sv.st2 = None
    if not (fw2 or fp):
        out = zk.lt_matmul(0, sv.o, m.out_proj.weight, m.out_proj.bias, x_in)
    else:
        sv.y, out = zk.lt_matmul(0, sv.o, m.out_proj.weight, m.out_proj.bias, act2="add", resid_b=x_in)
        if fw2:
            sv.st2 = zk.WhitenStats(sv.y, m.whiten2.num_groups)
    return out, sv


def _na_bwd(m, post, dec, sv, x_in, g, T, B):
    """-> (gradient w.r.t. the module input, dW0 (B,T,T) w.r.t. the head-0 weights)."""
    fb, fw1, fw2, fp = dec
    L, st = N.lib(), N.stream()
    gy = g
    if fp:
        gy = _balancer_bwd(post, sv.y, gy)
    if fw2:
        gy = _whiten_bwd(m.whiten2, sv.y, gy, sv.st2)
    _wgrad(m.out_proj.weight, m.out_proj.bias, gy, sv.o)
    do = zk.lt_matmul(1, gy, m.out_proj.weight)
    C = sv.u.shape[1] // 3
    dz = torch.empty_like(sv.z)
    du = torch.empty_like(sv.u)
    N.PROF[0] and N.profile_note("s2t_nonlin_out_bwd", 20.0 * T * B * C)
    N.check(L.s2t_nonlin_out_bwd(N.fp(do), N.fp(sv.z), N.fp(sv.u), T, B, C, N.fp(dz), N.fp(du), st),
            "nonlin_out_bwd")
    with zk.gemm_class(zk.CLS_D):
        dxs = zk.batched_matmul(2, sv.wm, dz)                                # W0^T @ dz, (B,T,C)
        dW0 = zk.batched_matmul(0, dz, sv.xs)                                # dz @ x^T, (B,T,T)
    N.PROF[0] and N.profile_note("s2t_nonlin_gate_bwd", 20.0 * T * B * C)
    N.check(L.s2t_nonlin_gate_bwd(N.fp(dxs), N.fp(sv.u), T, B, C, N.fp(du), st), "nonlin_gate_bwd")
    if fb:
        _balancer_bwd(m.balancer, sv.u[:, :C], du[:, :C], inplace=True)
    if fw1:
        du[:, C:2 * C] = _whiten_bwd(m.whiten1, sv.u[:, C:2 * C], du[:, C:2 * C].contiguous(),
                                     sv.st1)
    _wgrad(m.in_proj.weight, m.in_proj.bias, du, x_in)
    return zk.lt_matmul(1, du, m.in_proj.weight, None, g), dW0


def _attn_bwd_penalized(s, qkp3, pairs, dW0, H, qd, pd):
    """Rare branch: some raw score exceeded the limit of penalize_abs_values_gt, so its gradient
    term is non-zero.  The scores are rebuilt as an autograd graph (the materialised composition
    of zip_kernels.relpos_attention_weights, penalty included) and differentiated against the
    consumers' gradient w.r.t. W, materialised here from its factors."""
    T, B, _ = qkp3.shape
    dW = None
    for dO, v, _, dv in pairs:
        t = torch.matmul(dO.view(T, B, H, dv).permute(2, 1, 0, 3), v.view(T, B, H, dv).permute(2, 1, 3, 0))
        dW = t if dW is None else dW.add_(t)
    dW[0] += dW0
    with torch.enable_grad():
        q_ = qkp3.detach().requires_grad_(True)
        p_ = None if s.posp is None else s.posp.detach().requires_grad_(True)
        w = zk.relpos_attention_weights(
            q_, p_, H, qd, pd, None if s.a8 is None else s.a8.bool(),
            None if s.k8 is None else s.k8.bool(),
            penalize=lambda sc: S.penalize_abs_values_gt(sc, limit=PEN_LIMIT, penalty=PEN_VALUE))
        ins = [q_] if p_ is None else [q_, p_]
        gs = torch.autograd.grad(w, ins, dW)
    return gs[0].contiguous(), (None if p_ is None else gs[1].contiguous())


# ----------------------------------------------------------------------------- the layer
class _LayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, pos_emb, layer, chunk_size, d, a8, k8, fm):
        T, B, D = src.shape
        R = T * B
        dev = src.device
        L, st = N.lib(), N.stream()
        x0 = src.contiguous().view(R, D)
        sa = layer.self_attn_weights
        H, qd, pd = sa.num_heads, sa.query_head_dim, sa.pos_head_dim
        s = _Saved()
        s.d, s.dims, s.x0, s.a8, s.k8 = d, (T, B, D, H, qd, pd), x0, a8, k8

        # attention weights (reference zipformer.py:1966-2066)
        s.qkp = zk.lt_matmul(0, x0, sa.in_proj.weight, sa.in_proj.bias)
        s.kst = None
        if d.k_wh:
            s.kst = zk.WhitenStats(s.qkp[:, H * qd:2 * H * qd], sa.whiten_keys.num_groups)
        s.pos2 = s.posp = None
        if d.use_pos:
            s.pos2 = pos_emb.reshape(2 * T - 1, -1).contiguous().float()
            s.posp = zk.lt_matmul(0, s.pos2, sa.linear_pos.weight, None)
        W = torch.empty((H, B, T, T), dtype=_F32, device=dev)
        N.PROF[0] and N.profile_note("s2t_relpos_attn_fwd_flag" if d.penalize else "s2t_relpos_attn_fwd",
                                     4.0 * (s.qkp.numel() + W.numel()),
                                     2.0 * W.numel() * (qd + (pd if d.use_pos else 0)))
        s.pen_slot = s.pen_event = None
        if d.penalize:
            # the penalty has a gradient only where |score| > 25: the kernel raises a host-visible
            # flag if that happens at all, read (long after, no stall) when backward starts
            STATS["penalize"] += 1
            s.pen_slot = zk._pinned_slot()
            s.pen_slot[0] = 0.0
            N.check(L.s2t_relpos_attn_fwd_flag(N.fp(s.qkp), N.fp(s.posp), N.ptr(k8), N.ptr(a8), T,
                                               B, H, qd, pd, N.fp(W), PEN_LIMIT,
                                               ctypes.c_void_p(s.pen_slot.data_ptr()), st),
                    "s2t_relpos_attn_fwd_flag")
            s.pen_event = torch.cuda.Event()
            s.pen_event.record()
        else:
            N.check(L.s2t_relpos_attn_fwd(N.fp(s.qkp), N.fp(s.posp), N.ptr(k8), N.ptr(a8), T, B, H,
                                          qd, pd, N.fp(W), st), "s2t_relpos_attn_fwd")
        s.W = W

        x1, s.ff1 = _ff_fwd(layer.feed_forward1, d.ff1, x0)
        x2, s.na = _na_fwd(layer.nonlin_attention, d.na, x1, W, T, B)
        x3, s.sa1 = _sa_fwd(layer.self_attn1, d.sa1, x2, W, T, B, H)
        x4, s.cv1 = _conv_fwd(layer.conv_module1, d.cv1, x3, T, B, chunk_size, k8)
        x5, s.ff2 = _ff_fwd(layer.feed_forward2, d.ff2, x4)
        x6 = _e(R, D, dev)
        N.PROF[0] and N.profile_note("s2t_bypass_fwd", 12.0 * R * D)
        N.check(L.s2t_bypass_fwd(N.fp(x0), N.fp(x5), N.fp(layer.bypass_mid.bypass_scale), R, D,
                                 N.fp(x6), st), "s2t_bypass_fwd")
        x7, s.sa2 = _sa_fwd(layer.self_attn2, d.sa2, x6, W, T, B, H)
        x8, s.cv2 = _conv_fwd(layer.conv_module2, d.cv2, x7, T, B, chunk_size, k8)
        x9, s.ff3 = _ff_fwd(layer.feed_forward3, d.ff3, x8)
        norm = layer.norm
        s.nscales = torch.empty(R, dtype=_F32, device=dev)
        x11 = _e(R, D, dev)
        # BiasNorm + the layer's bypass in one pass (norm(x9) is never stored); the stack's feature
        # mask rides in it unless a gradient-shaping op of this call needs the unmasked output
        s.fm, s.fm_fused = fm, fm is not None and not (d.wh_out or d.bal2)
        N.PROF[0] and N.profile_note("s2t_norm_bypass_fwd", 12.0 * R * D)
        N.check(L.s2t_norm_bypass_fwd(N.fp(x9), N.fp(norm.bias), ctypes.c_void_p(norm.log_scale.data_ptr()),
                                      N.fp(x0), N.fp(layer.bypass.bypass_scale),
                                      N.fp(fm) if s.fm_fused else None, B, R, D, N.fp(x11),
                                      N.fp(s.nscales), st), "s2t_norm_bypass_fwd")
        s.wst = zk.WhitenStats(x11, layer.whiten.num_groups) if d.wh_out else None
        s.x = (x1, x2, x3, x4, x5, x6, x7, x8, x9, None, x11)
        ctx.s, ctx.layer = s, layer
        out = x11.view(T, B, D)
        return out * fm if (fm is not None and not s.fm_fused) else out

    @staticmethod
    def backward(ctx, g):
        s, layer = ctx.s, ctx.layer
        ctx.s = None
        _PEND.clear()
        d = s.d
        T, B, D, H, qd, pd = s.dims
        R = T * B
        dev = g.device
        L, st = N.lib(), N.stream()
        x0 = s.x0
        x1, x2, x3, x4, x5, x6, x7, x8, x9, x10, x11 = s.x
        g = g.contiguous().view(R, D)
        if g.dtype != _F32:
            g = g.float()
        if g.data_ptr() % 16:
            g = g.clone()
        if s.fm is not None and not s.fm_fused:
            g = (g.view(T, B, D) * s.fm).view(R, D)
        if d.wh_out:
            g = _whiten_bwd(layer.whiten, x11, g, s.wst)
        if d.bal2:
            g = _balancer_bwd(layer.balancer2, x11, g)
        # per-channel parameter gradients: [bypass scale | bypass_mid scale | norm bias | log_scale]
        acc = _layer_acc(dev, D)
        off = lambda n: ctypes.c_void_p(acc.data_ptr() + 4 * n)      # noqa: E731

        byp = layer.bypass
        norm = layer.norm
        d0 = _e(R, D, dev)
        g9 = _e(R, D, dev)
        N.PROF[0] and N.profile_note("s2t_norm_bypass_bwd", 20.0 * R * D)
        N.check(L.s2t_norm_bypass_bwd(N.fp(x9), N.fp(norm.bias), N.fp(s.nscales), N.fp(x0),
                                      N.fp(byp.bypass_scale), N.fp(g), N.fp(s.fm) if s.fm_fused else None,
                                      B, R, D, N.fp(g9), N.fp(d0), off(0), off(2 * D), off(3 * D), st),
                "s2t_norm_bypass_bwd")
        if d.bal1:
            g9 = _balancer_bwd(layer.balancer1, x9, g9)

        pairs = []
        g8 = _ff_bwd(layer.feed_forward3, layer.balancer_ff3, d.ff3, s.ff3, x8, g9)
        g7 = _conv_bwd(layer.conv_module2, d.cv2, s.cv2, x7, g8, T, B, s.k8)
        g6 = _sa_bwd(layer.self_attn2, d.sa2, s.sa2, x6, g7, s.W, T, B, H, pairs)

        mid = layer.bypass_mid
        d0m = _e(R, D, dev)
        g5 = _e(R, D, dev)
        N.PROF[0] and N.profile_note("s2t_bypass_bwd_acc", 24.0 * R * D)
        N.check(L.s2t_bypass_bwd_acc(N.fp(x0), N.fp(x5), N.fp(mid.bypass_scale), N.fp(g6),
                                     N.fp(d0), R, D, N.fp(d0m), N.fp(g5), off(D), st),
                "s2t_bypass_bwd_acc")
        _commit([(byp.bypass_scale, acc[:D], byp.scale_min, byp.scale_max, d.byp_lim),
                 (mid.bypass_scale, acc[D:2 * D], mid.scale_min, mid.scale_max, d.mid_lim),
                 (norm.bias, acc[2 * D:3 * D], 0.0, 0.0, False),
                 (norm.log_scale, acc[3 * D:3 * D + 1], norm.log_scale_min, norm.log_scale_max,
                  d.norm_lim)])

        g4 = _ff_bwd(layer.feed_forward2, layer.balancer_ff2, d.ff2, s.ff2, x4, g5)
        g3 = _conv_bwd(layer.conv_module1, d.cv1, s.cv1, x3, g4, T, B, s.k8)
        g2 = _sa_bwd(layer.self_attn1, d.sa1, s.sa1, x2, g3, s.W, T, B, H, pairs)
        g1, dW0 = _na_bwd(layer.nonlin_attention, layer.balancer_na, d.na, s.na, x1, g2, T, B)
        g0 = _ff_bwd(layer.feed_forward1, None, d.ff1, s.ff1, x0, g1)

        # attention weights: delta from the consumers, then dS -> dq, dk, dp, dpos
        sa = layer.self_attn_weights
        delta = torch.empty((H, B, T), dtype=_F32, device=dev)
        (dO1, _, O1, dv1), (dO2, _, O2, dv2) = pairs
        N.PROF[0] and N.profile_note("s2t_attn_delta_pairs", 4.0 * (s.W.numel() // H + dW0.numel() + 2 * (dO1.numel()
                                                        + dO2.numel()) + delta.numel()))
        N.check(L.s2t_attn_delta_pairs(N.fp(s.W), N.fp(dW0), N.fp(dO1), N.fp(O1), dv1, N.fp(dO2),
                                       N.fp(O2), dv2, T, B, H, N.fp(delta), st),
                "s2t_attn_delta_pairs")
        qkp3 = s.qkp.view(T, B, -1)
        pen_active = False
        if d.penalize:
            s.pen_event.synchronize()
            pen_active = float(s.pen_slot[0]) != 0.0
        if pen_active:
            STATS["penalty_active"] += 1
            dqkp, dpos = _attn_bwd_penalized(s, qkp3, pairs, dW0, H, qd, pd)
        else:
            dqkp, dpos = zk._attn_bwd_call(qkp3, s.posp, s.k8, s.a8, H, qd, pd, s.W, None, dW0,
                                           pairs, delta)
        dqkp = dqkp.view(R, -1)
        if d.k_wh or d.k_bal:
            ks = slice(H * qd, 2 * H * qd)
            gk = dqkp[:, ks].contiguous()
            if d.k_wh:
                gk = _whiten_bwd(sa.whiten_keys, s.qkp[:, ks], gk, s.kst)
            if d.k_bal:
                gk = _balancer_bwd(sa.balance_keys, s.qkp[:, ks], gk)
            dqkp[:, ks] = gk
        if dpos is not None:
            _wgrad(sa.linear_pos.weight, None, dpos, s.pos2)
        _wgrad(sa.in_proj.weight, sa.in_proj.bias, dqkp, x0)
        gx = zk.lt_matmul(1, dqkp, sa.in_proj.weight, None, g0, resid_b=d0m)
        pend = list(_PEND)
        _PEND.clear()
        zk.wgrad_group(pend)
        return gx.view(T, B, D), None, None, None, None, None, None, None
