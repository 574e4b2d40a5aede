"""Host side of the native per-layer executor (csrc/zip_layer.hip).

`zip_layer.run` draws a layer call's random decisions (reference call sites and order:
model/encoder/zipformer.py:909-1338, model/layer/scaling.py:836, 1071, 1186) and, when this
module can serve the call, hands them -- with a descriptor of the layer and one device
workspace -- to `s2t_zip_layer_fwd` / `s2t_zip_layer_bwd`: two C calls per layer and step instead
of ~105 ctypes calls, ~60 tensor allocations and the Python between them.  `zip_layer._LayerFn`
stays the reference implementation of the same launch sequence and serves what this path refuses:
shapes whose GEMM plan has not been timed yet (the first step), profiling of every entry point,
the off-by-default statistics variants.
"""
import ctypes
import os

import torch

from . import _native as N
from . import flat
from . import planes
from . import zip_kernels as zk

ENABLED = os.environ.get("S2T_LAYER_NATIVE", "1") == "1"
_BAL_FWD_SIDE = os.environ.get("S2T_BAL_FWD_SIDE", "1") == "1"   # Balancer column statistics in forward, side stream
CALLS = [0, 0]           # forward / backward calls served natively (tests assert the path really ran)
_F32 = torch.float32
NDEC, NWHITEN = 32, 11

c_fp = ctypes.c_void_p


class Lin(ctypes.Structure):
    _fields_ = [("w", c_fp), ("b", c_fp), ("gw", c_fp), ("gb", c_fp), ("pf", c_fp), ("pb", c_fp),
                ("N", ctypes.c_int), ("K", ctypes.c_int)]


class Bal(ctypes.Structure):
    _fields_ = [("min_mean", ctypes.c_float), ("max_mean", ctypes.c_float), ("min_rms", ctypes.c_float),
                ("max_rms", ctypes.c_float), ("grad_scale", ctypes.c_float)]


class Wh(ctypes.Structure):
    _fields_ = [("groups", ctypes.c_int), ("limit", ctypes.c_float), ("grad_scale", ctypes.c_float)]


class Ff(ctypes.Structure):
    _fields_ = [("in_", Lin), ("out", Lin), ("hidden", Bal), ("out_wh", Wh), ("post", Bal)]


class Sa(ctypes.Structure):
    _fields_ = [("in_", Lin), ("out", Lin), ("wh", Wh)]


class Conv(ctypes.Structure):
    _fields_ = [("in_", Lin), ("out", Lin), ("bal1", Bal), ("bal2", Bal), ("wh", Wh),
                ("K", ctypes.c_int), ("causal", ctypes.c_int),
                ("wc", c_fp), ("bc", c_fp), ("wk", c_fp), ("bk", c_fp), ("scale", c_fp),
                ("gwc", c_fp), ("gbc", c_fp), ("gwk", c_fp), ("gbk", c_fp), ("gscale", c_fp)]


class Na(ctypes.Structure):
    _fields_ = [("in_", Lin), ("out", Lin), ("bal", Bal), ("wh1", Wh), ("wh2", Wh), ("post", Bal)]


class Param(ctypes.Structure):
    _fields_ = [("x", c_fp), ("grad", c_fp), ("lo", ctypes.c_float), ("hi", ctypes.c_float)]


class Desc(ctypes.Structure):
    """Mirror of S2tZipLayerDesc (include/s2t_mi355.h)."""
    _fields_ = [("D", ctypes.c_int), ("H", ctypes.c_int), ("qd", ctypes.c_int), ("pd", ctypes.c_int),
                ("pos_dim", ctypes.c_int),
                ("attn_in", Lin), ("attn_pos", Lin), ("bal_keys", Bal), ("wh_keys", Wh),
                ("ff", Ff * 3), ("na", Na), ("sa", Sa * 2), ("cv", Conv * 2),
                ("byp_mid", Param), ("byp", Param), ("norm_bias", Param), ("norm_ls", Param),
                ("bal1", Bal), ("bal2", Bal), ("wh_out", Wh)]


class WhScratch(ctypes.Structure):
    _fields_ = [("C", ctypes.c_int), ("acc", c_fp), ("ws", c_fp), ("tab", c_fp), ("buf", c_fp),
                ("blocks", ctypes.c_int)]


class Call(ctypes.Structure):
    """Mirror of S2tZipLayerCall."""
    _fields_ = [("T", ctypes.c_int), ("B", ctypes.c_int), ("chunk_size", ctypes.c_int),
                ("x0", c_fp), ("pos", c_fp), ("k8", c_fp), ("a8", c_fp), ("fm", c_fp),
                ("out", c_fp), ("g", c_fp), ("gx", c_fp),
                ("dec", ctypes.c_int * NDEC),
                ("bal_ws", c_fp), ("layer_acc", c_fp), ("wh", WhScratch * 8), ("nwh", ctypes.c_int),
                ("lt_ws", c_fp), ("lt_ws_bytes", ctypes.c_long),
                ("x3p_on", ctypes.c_int), ("x3p_tile", ctypes.c_int), ("x3p_margin", ctypes.c_float),
                ("whiten_x3p", ctypes.c_int),
                ("conv_w_side", ctypes.c_int), ("conv_fused", ctypes.c_int), ("stats_side", ctypes.c_int),
                ("wgrad_side", ctypes.c_int), ("bmm_own", ctypes.c_int), ("bal_epi", ctypes.c_int),
                ("whiten_sq", ctypes.c_int), ("bal_fwd_side", ctypes.c_int), ("whiten_fwd_pg", ctypes.c_int)]


def _dp(t):
    return None if t is None else t.data_ptr()


def _lin(dst, mod_w, mod_b):
    """Fill a Lin from a weight (N,K) / bias that live in a FlatStore; False if they do not."""
    w = mod_w
    if not (flat.owned(w) and w.grad is not None and w.dim() == 2 and w.is_contiguous()):
        return False
    if mod_b is not None and not (flat.owned(mod_b) and mod_b.grad is not None):
        return False
    dst.w, dst.gw = w.data_ptr(), w.grad.data_ptr()
    dst.b, dst.gb = _dp(mod_b), (None if mod_b is None else mod_b.grad.data_ptr())
    dst.N, dst.K = int(w.shape[0]), int(w.shape[1])
    dst.pf, dst.pb = planes.pieces(w, 0), planes.pieces(w, 1)
    return True


def _bal(dst, m):
    dst.min_mean, dst.max_mean, dst.min_rms, dst.max_rms, dst.grad_scale = m.cfg(2)[:5]
    return m.num_channels <= 1024


def _wh(dst, m):
    dst.groups, dst.limit, dst.grad_scale = int(m.num_groups), float(m.whitening_limit), float(m.grad_scale)


def _param(dst, p, lo=0.0, hi=0.0):
    if not (flat.owned(p) and p.grad is not None):
        return False
    dst.x, dst.grad, dst.lo, dst.hi = p.data_ptr(), p.grad.data_ptr(), float(lo), float(hi)
    return True


class _Layer:
    """Everything static about one layer object: descriptor, module lists, workspace sizes."""
    __slots__ = ("desc", "ok", "whitens", "params", "store", "weights", "sizes", "key", "versions")


def _build(layer):
    sa = layer.self_attn_weights
    d = Desc()
    ok = True
    d.D, d.H, d.qd, d.pd = layer.embed_dim, sa.num_heads, sa.query_head_dim, sa.pos_head_dim
    d.pos_dim = int(sa.linear_pos.weight.shape[1])
    weights = []

    def lin(dst, m):
        nonlocal ok
        ok = _lin(dst, m.weight, m.bias) and ok
        weights.append(m.weight)

    lin(d.attn_in, sa.in_proj)
    lin(d.attn_pos, sa.linear_pos)
    ok = _bal(d.bal_keys, sa.balance_keys) and ok
    _wh(d.wh_keys, sa.whiten_keys)
    for i, (m, post) in enumerate(((layer.feed_forward1, None), (layer.feed_forward2, layer.balancer_ff2),
                                   (layer.feed_forward3, layer.balancer_ff3))):
        f = d.ff[i]
        lin(f.in_, m.in_proj)
        lin(f.out, m.out_proj)
        ok = _bal(f.hidden, m.hidden_balancer) and ok
        _wh(f.out_wh, m.out_whiten)
        if post is not None:
            ok = _bal(f.post, post) and ok
    na = layer.nonlin_attention
    lin(d.na.in_, na.in_proj)
    lin(d.na.out, na.out_proj)
    ok = _bal(d.na.bal, na.balancer) and ok
    _wh(d.na.wh1, na.whiten1)
    _wh(d.na.wh2, na.whiten2)
    ok = _bal(d.na.post, layer.balancer_na) and ok
    for i, m in enumerate((layer.self_attn1, layer.self_attn2)):
        lin(d.sa[i].in_, m.in_proj)
        lin(d.sa[i].out, m.out_proj)
        _wh(d.sa[i].wh, m.whiten)
    for i, m in enumerate((layer.conv_module1, layer.conv_module2)):
        cv = d.cv[i]
        lin(cv.in_, m.in_proj)
        lin(cv.out, m.out_proj)
        ok = _bal(cv.bal1, m.balancer1) and ok
        ok = _bal(cv.bal2, m.balancer2) and ok
        _wh(cv.wh, m.whiten)
        dw = m.depthwise_conv
        if isinstance(dw, torch.nn.Conv1d):
            ok = False
            continue
        cv.K, cv.causal = int(dw.kernel_size), int(bool(m.causal))
        plist = (dw.causal_conv.weight, dw.causal_conv.bias, dw.chunkwise_conv.weight,
                 dw.chunkwise_conv.bias, dw.chunkwise_conv_scale)
        grads = zk.direct_grads(plist)
        if grads is None:
            ok = False
            continue
        cv.wc, cv.bc, cv.wk, cv.bk, cv.scale = (_dp(p) for p in plist)
        cv.gwc, cv.gbc, cv.gwk, cv.gbk, cv.gscale = (_dp(g) for g in grads)
    mid, byp, norm = layer.bypass_mid, layer.bypass, layer.norm
    ok = _param(d.byp_mid, mid.bypass_scale, mid.scale_min, mid.scale_max) and ok
    ok = _param(d.byp, byp.bypass_scale, byp.scale_min, byp.scale_max) and ok
    ok = _param(d.norm_bias, norm.bias) and ok
    ok = _param(d.norm_ls, norm.log_scale, norm.log_scale_min, norm.log_scale_max) and ok
    ok = _bal(d.bal1, layer.balancer1) and ok
    ok = _bal(d.bal2, layer.balancer2) and ok
    _wh(d.wh_out, layer.whiten)
    L = _Layer()
    L.desc, L.ok = d, ok
    L.whitens = [sa.whiten_keys, layer.feed_forward1.out_whiten, na.whiten1, na.whiten2,
                 layer.self_attn1.whiten, layer.conv_module1.whiten, layer.feed_forward2.out_whiten,
                 layer.self_attn2.whiten, layer.conv_module2.whiten, layer.feed_forward3.out_whiten,
                 layer.whiten]
    # parameters whose gradient this executor writes (exactly the ones zip_layer._LayerFn reports):
    # the data-parallel reducer is told once per parameter and call
    lins = [sa.in_proj, sa.linear_pos]
    for m in (layer.feed_forward1, layer.feed_forward2, layer.feed_forward3, na, layer.self_attn1,
              layer.self_attn2, layer.conv_module1, layer.conv_module2):
        lins += [m.in_proj, m.out_proj]
    ps = []
    for m in lins:
        ps.append(m.weight)
        if m.bias is not None:
            ps.append(m.bias)
    for m in (layer.conv_module1, layer.conv_module2):
        dw = m.depthwise_conv
        if not isinstance(dw, torch.nn.Conv1d):
            ps += [p for p in (dw.causal_conv.weight, dw.causal_conv.bias, dw.chunkwise_conv.weight,
                               dw.chunkwise_conv.bias, dw.chunkwise_conv_scale) if p is not None]
    ps += [byp.bypass_scale, mid.bypass_scale, norm.bias, norm.log_scale]
    L.params = ps
    ent = flat._OWNER.get(id(mid.bypass_scale))
    L.store = ent[0] if ent is not None else None
    L.weights = weights
    L.versions = None
    L.sizes = {}
    L.key = (id(L.store), None if L.store is None else L.store.flat_p.data_ptr())
    return L


def _static(layer):
    L = layer.__dict__.get("_zn_layer")
    if L is not None:
        ent = flat._OWNER.get(id(layer.bypass_mid.bypass_scale))
        st = ent[0] if ent is not None else None
        if L.key != (id(st), None if st is None else st.flat_p.data_ptr()):
            L = None                                   # the model moved to another FlatStore
    if L is None:
        L = _build(layer)
        layer.__dict__["_zn_layer"] = L
    return L


_DEC_ORDER = None


def _decisions(d):
    """zip_layer._Plan -> the dec[] vector (indices: include/s2t_mi355.h)."""
    v = [0] * NDEC
    v[0], v[1], v[2], v[3] = d.k_bal, d.k_wh, d.use_pos, d.penalize
    v[4:7] = d.ff1
    v[7:11] = d.na
    v[11] = d.sa1
    v[12:15] = d.cv1
    v[15:18] = d.ff2
    v[18] = d.mid_lim
    v[19] = d.sa2
    v[20:23] = d.cv2
    v[23:26] = d.ff3
    v[26], v[27], v[28], v[29], v[30] = d.bal1, d.norm_lim, d.byp_lim, d.bal2, d.wh_out
    return [int(bool(x)) for x in v]


_ADHOC_OK = {}


def _wh_scratch(dst, dev, C):
    acc, ws = zk._whiten_scratch(dev, C)
    dst.C, dst.acc, dst.ws = C, acc.data_ptr(), ws.data_ptr()
    ent = planes.adhoc_entry(C, C, 1, dev)
    if ent is None:
        dst.tab = dst.buf = None
        dst.blocks = 0
    else:
        dst.tab, dst.buf, dst.blocks = ent[0].data_ptr(), ent[1].data_ptr(), ent[2]


def _side_handle():
    if not zk._Side.enabled:
        return None
    if zk._Side.handle is None:
        h = N.lib().s2t_side_stream()
        if not h:
            zk._Side.enabled = False
            return None
        zk._Side.handle = ctypes.c_void_p(h)
    return zk._Side.handle


def usable(layer, T, B):
    """Can the native executor serve this layer call?  (zip_layer.eligible has passed already.)"""
    if not ENABLED or "S2T_ATTN_FWD_OLD" in os.environ:
        return None
    # a profile of an entry point whose launches are only seen from the Python call sites ("*", or any
    # single entry that is not sampled inside the library) needs the Python executor
    if N._Prof.target is not None and N._Prof.target not in N.KERNEL_TIMED:
        return None
    L = _static(layer)
    if not L.ok:
        return None
    # weight pieces current?  The first lookup refreshes ALL of them when the store's epoch moved
    # (the optimizers, the DP broadcast); an in-place torch edit of any single matrix of the layer
    # (partial load_state_dict, weight.mul_) shows in that parameter's _version: one integer
    # compare per weight, and a refresh through the owning arena when one moved
    planes.pieces(L.weights[0], 0)
    if L.versions is None:                         # (weight, its arena entry) pairs, once per layer object
        L.versions = []
        for w in L.weights:
            a = planes.arena_of(w)
            e = None if a is None else a.entries.get(w.data_ptr())
            if e is not None:
                L.versions.append((w, e, a))
    for w, e, a in L.versions:
        if e.version != w._version:
            a.refresh()
            planes._HOT.clear()
            break
    if N.lib().s2t_zip_layer_plans_missing(ctypes.byref(L.desc), T, B) != 0:
        return None
    return L


def _fill_call(L, T, B, D, chunk_size, x0, pos2, a8, k8, fm, dec, dev):
    c = Call()
    c.T, c.B, c.chunk_size = T, B, int(chunk_size)
    c.x0, c.pos, c.k8, c.a8, c.fm = x0.data_ptr(), _dp(pos2), _dp(k8), _dp(a8), _dp(fm)
    for i, v in enumerate(dec):
        c.dec[i] = v
    c.bal_ws = zk._balancer_workspace(dev)[0].data_ptr()
    from . import zip_layer as zl
    c.layer_acc = zl._layer_acc(dev, D).data_ptr()
    d = L.desc
    cs = []
    for C in (d.H * d.qd, D, d.na.in_.N // 3):
        if C not in cs:
            cs.append(C)
    for i, C in enumerate(cs):
        _wh_scratch(c.wh[i], dev, C)
    c.nwh = len(cs)
    ws = zk._lt_workspace(dev)
    c.lt_ws, c.lt_ws_bytes = ws.data_ptr(), ws.numel()
    c.x3p_on, c.x3p_tile, c.x3p_margin = int(zk.X3P["on"]), int(zk.X3P["tile"]), float(zk.X3P["margin"])
    c.whiten_x3p = int(zk._WHITEN_X3P)
    side = zk._Side.enabled
    c.conv_w_side = int(zk._CONV_W_SIDE and side)
    c.conv_fused = 0                      # (the one-kernel conv backward: removed in round 5)
    c.stats_side = int(zk._STATS_SIDE and side)
    c.wgrad_side = int(side)
    c.bmm_own = int(zk._BMM_OWN)
    c.bal_epi = int(zk._BAL_EPI)
    c.whiten_sq = int(zk._WHITEN_SQ)
    c.bal_fwd_side = int(_BAL_FWD_SIDE and side)
    c.whiten_fwd_pg = int(zk._WHITEN_FWD_PG)
    return c


def _err(rc, what):
    msg = ctypes.cast(N.lib().s2t_zip_layer_error(), ctypes.c_char_p).value
    raise RuntimeError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")


def _ws_floats(L, call, T, B, backward):
    key = (T, B, int(call.chunk_size >= 0), backward, call.fm is not None)
    n = L.sizes.get(key)
    if n is None:
        probe = Call()
        ctypes.memmove(ctypes.byref(probe), ctypes.byref(call), ctypes.sizeof(Call))
        for i in range(NDEC):
            probe.dec[i] = 1
        n = int(N.lib().s2t_zip_layer_ws_floats(ctypes.byref(L.desc), ctypes.byref(probe), backward))
        if n <= 0:
            raise RuntimeError("s2t_zip_layer_ws_floats failed")
        L.sizes[key] = n
    return n


_STATE_BYTES = [0]


class _NativeLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, pos_emb, layer, L, chunk_size, d, a8, k8, fm):
        from . import zip_layer as zl
        T, B, D = src.shape
        dev = src.device
        x0 = src.contiguous().view(T * B, D)
        if x0.data_ptr() % 16:
            x0 = x0.clone()
        pos2 = pos_emb.reshape(2 * T - 1, -1).contiguous().float() if d.use_pos else None
        dec = _decisions(d)
        call = _fill_call(L, T, B, D, chunk_size, x0, pos2, a8, k8, fm, dec, dev)
        x11 = torch.empty((T * B, D), dtype=_F32, device=dev)
        call.out = x11.data_ptr()
        n = _ws_floats(L, call, T, B, 0)
        wsf = torch.empty(n, dtype=_F32, device=dev)
        if not _STATE_BYTES[0]:
            _STATE_BYTES[0] = int(N.lib().s2t_zip_layer_state_bytes())
        state = ctypes.create_string_buffer(_STATE_BYTES[0])
        if d.penalize:
            zl.STATS["penalize"] += 1
        rc = N.lib().s2t_zip_layer_fwd(ctypes.byref(L.desc), ctypes.byref(call), state, wsf.data_ptr(), n,
                                       N.stream(), _side_handle())
        if rc != 0:
            _err(rc, "s2t_zip_layer_fwd")
        if call.bal_fwd_side or (call.stats_side and (d.k_wh or d.ff1[1] or d.na[1] or d.na[2] or d.sa1 or d.cv1[2]
                                                      or d.ff2[1] or d.sa2 or d.cv2[2] or d.ff3[1] or d.wh_out)):
            # the side stream may still be reading the workspace / the output when this object dies
            # without a backward pass (a forward under train()): keep them until the join
            zk._Side.keep.append((wsf, x11, x0))
        CALLS[0] += 1
        ctx.pack = (layer, L, call, state, wsf, x0, pos2, a8, k8, fm, x11, d, (T, B, D))
        fm_fused = fm is not None and not (d.wh_out or d.bal2)
        out = x11.view(T, B, D)
        return out * fm if (fm is not None and not fm_fused) else out

    @staticmethod
    def backward(ctx, g):
        from . import zip_layer as zl
        layer, L, call, state, wsf, x0, pos2, a8, k8, fm, out, d, (T, B, D) = ctx.pack
        ctx.pack = None
        R = T * B
        dev = g.device
        g = g.contiguous().view(R, D)
        if g.dtype != _F32:
            g = g.float()
        if g.data_ptr() % 16:
            g = g.clone()
        fm_fused = fm is not None and not (d.wh_out or d.bal2)
        if fm is not None and not fm_fused:
            g = (g.view(T, B, D) * fm).view(R, D)
        gx = torch.empty((T, B, D), dtype=_F32, device=dev)
        call.g, call.gx = g.data_ptr(), gx.data_ptr()
        n = _ws_floats(L, call, T, B, 1)
        wsb = torch.empty(n, dtype=_F32, device=dev)
        # registers the end-of-backward join of the side stream and keeps the operands of its
        # kernels (weight gradients, conv parameter gradients) alive until then
        # (k8 / a8 too: the stack's masks die with its first layer's backward, and the conv modules'
        #  parameter-gradient kernels -- which read the padding mask -- leave at the END of this call)
        side = zk._side_launch_stream(wsb, wsf, g, x0, out, pos2, k8, a8) if zk._Side.enabled else None
        lib = N.lib()
        rc = lib.s2t_zip_layer_bwd(ctypes.byref(L.desc), ctypes.byref(call), state, wsb.data_ptr(), n, 0,
                                   N.stream(), side)
        if rc == 1:
            # rare: a raw score exceeded the limit of penalize_abs_values_gt on a call that drew the
            # penalty -- the attention-weights backward through the materialised autograd graph
            zl.STATS["penalty_active"] += 1
            _penalized(layer, L, state, wsf, wsb, T, B, a8, k8, d)
            rc = lib.s2t_zip_layer_bwd(ctypes.byref(L.desc), ctypes.byref(call), state, wsb.data_ptr(), n, 2,
                                       N.stream(), side)
        if rc != 0:
            _err(rc, "s2t_zip_layer_bwd")
        for site, m in enumerate(L.whitens):
            a = lib.s2t_zip_layer_info(state, 0, site)
            if a >= 0:
                m.prob = m.max_prob if a else m.min_prob
        st = L.store
        if st is not None and st.on_grad is not None:
            skip = None if d.use_pos else layer.self_attn_weights.linear_pos.weight
            for p in L.params:
                if p is not skip:
                    flat.grad_written(p)
        CALLS[1] += 1
        return gx, None, None, None, None, None, None, None, None


def _view(ws_list, addr, shape):
    """A tensor over `shape` floats at device address `addr` inside one of the workspaces."""
    n = 1
    for s in shape:
        n *= s
    for ws in ws_list:
        off = (addr - ws.data_ptr()) // 4
        if 0 <= off and off + n <= ws.numel():
            return ws[off:off + n].view(shape)
    raise RuntimeError("zip_native: address outside the workspaces")


def _penalized(layer, L, state, wsf, wsb, T, B, a8, k8, d):
    from . import zip_layer as zl
    lib = N.lib()
    desc = L.desc
    H, qd, pd = desc.H, desc.qd, desc.pd
    Dp = desc.attn_in.N
    info = lambda w, i=0: int(lib.s2t_zip_layer_info(state, w, i))      # noqa: E731
    wss = (wsf, wsb)
    qkp3 = _view(wss, info(2), (T, B, Dp))
    posp = _view(wss, info(3), (2 * T - 1, H * pd)) if info(3) else None
    pairs = []
    for k in (0, 1):
        dv = info(10, k)
        pairs.append((_view(wss, info(5, k), (T * B, H * dv)), _view(wss, info(6, k), (T * B, H * dv)), None, dv))
    dW0 = _view(wss, info(7), (B, T, T))
    s = zl._Saved()
    s.posp, s.a8, s.k8 = posp, a8, k8
    dqkp, dpos = zl._attn_bwd_penalized(s, qkp3, pairs, dW0, H, qd, pd)
    _view(wss, info(8), (T, B, Dp)).copy_(dqkp.view(T, B, Dp))
    if dpos is not None:
        _view(wss, info(9), (2 * T - 1, H * pd)).copy_(dpos)


def run(layer, L, src, pos_emb, chunk_size, d, a8, k8, fm):
    return _NativeLayerFn.apply(src, pos_emb, layer, L, chunk_size, d, a8, k8, fm)
