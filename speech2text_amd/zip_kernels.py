"""Zipformer hot-op entry points used by speech2text_amd.model.* (the seam to the HIP ABI).

Each function is the single place a given fused op is launched from.  Ops are moved from
torch-op compositions (rocBLAS GEMMs + elementwise) to hand-written gfx950 kernels one at a
time; see DESIGN.md for the status table of each.
"""
import ctypes
import os

import torch
import torch.nn.functional as F

from . import _native as N
from . import flat
from . import planes

_SW = {True: (4.0, 0.035), False: (1.0, 0.313261687)}   # SwooshL / SwooshR (offset, constant)


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("speech2text_amd zipformer kernels run on the GPU only "
                               "(no CPU fallback); got a CPU tensor")


def _c16(t):
    """contiguous + 16-byte aligned (the streaming kernels use 16-byte lane accesses)."""
    t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone()
    return t


# ------------------------------------------------------------------ swoosh
class _Swoosh(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, is_l):
        ctx.save_for_backward(x)
        ctx.is_l = is_l
        return swoosh_forward(x, is_l)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return swoosh_backward(x, g, ctx.is_l), None


def swoosh_forward(x, is_l):
    """log(1+exp(x-off)) - 0.08x - c   (HIP: zip_elem.hip swoosh_fwd_kernel)"""
    _dev(x)
    off, c = _SW[is_l]
    x = _c16(x.float())
    y = torch.empty_like(x)
    N.PROF[0] and N.profile_note("s2t_swoosh_fwd", 8.0 * x.numel())
    N.check(N.lib().s2t_swoosh_fwd(N.fp(x), N.fp(y), x.numel(), off, c, N.stream()), "swoosh")
    return y


def swoosh_backward(x, g, is_l, mask=None):
    """g * (sigmoid(x - off) - 0.08) [* mask]   (HIP: swoosh_bwd_kernel)"""
    _dev(x, g)
    off, _ = _SW[is_l]
    x = _c16(x.float())
    g = _c16(g.float())
    d = torch.empty_like(x)
    N.PROF[0] and N.profile_note("s2t_swoosh_bwd", 12.0 * x.numel())
    N.check(N.lib().s2t_swoosh_bwd(N.fp(x), N.fp(g), N.fp(d), x.numel(), off, N.stream()),
            "swoosh_bwd")
    return d if mask is None else d * mask


def swoosh(x, is_l):
    return _Swoosh.apply(x, is_l)


class _SwooshLinear(torch.autograd.Function):
    """y = linear(swoosh(x) [* mask], W, b); only x is saved (reference scaling.py:1512-1583).
    Backward: the weight / bias gradients come from the TN MFMA GEMM with the swoosh applied to
    x while it is staged (no recomputed activation tensor) and are accumulated straight into
    the flat gradient buffer."""

    @staticmethod
    def forward(ctx, x, weight, bias, is_l, mask, residual):
        ctx.save_for_backward(x, weight, mask)
        ctx.is_l = is_l
        ctx.params = (weight, bias)
        ctx.has_res = residual is not None
        h = swoosh_forward(x, is_l)
        if mask is not None:
            h = h * mask
        r2 = None if residual is None else _rows(residual)
        y = lt_matmul(0, _rows(h), weight, bias, r2)
        return y.view(x.shape[:-1] + (weight.shape[0],))

    @staticmethod
    def backward(ctx, g):
        x, weight, mask = ctx.saved_tensors
        wp, bp = ctx.params
        g2 = _rows(g)
        x2 = x.reshape(-1, x.shape[-1])
        if mask is None and wgrad_into(wp, bp, g2, x2, pro=1 if ctx.is_l else 2):
            dw = db = None
        else:
            h = swoosh_forward(x, ctx.is_l)
            if mask is not None:
                h = h * mask
            dw, db = linear_wgrad(g2, h.reshape(-1, h.shape[-1]), bp is not None)
        dh = lt_matmul(1, g2, weight).view(x.shape)
        dx = swoosh_backward(x, dh, ctx.is_l, mask)
        return dx, dw, db, None, None, (g if ctx.has_res else None)


def swoosh_linear(x, weight, bias, is_l, mask=None, residual=None):
    return _SwooshLinear.apply(x, weight, bias, is_l, mask, residual)


# ------------------------------------------------------------------ BiasNorm
class _BiasNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, log_scale):
        _dev(x, bias, log_scale)
        D = x.shape[-1]
        x = x.contiguous().float()
        rows = x.numel() // D
        y = torch.empty_like(x)
        scales = torch.empty(rows, dtype=torch.float32, device=x.device)
        bias = bias.contiguous().float()
        N.PROF[0] and N.profile_note("s2t_biasnorm_fwd", 8.0 * x.numel())
        N.check(N.lib().s2t_biasnorm_fwd(N.fp(x), N.fp(bias),
                                         N.fp(log_scale.reshape(1).contiguous().float()), rows, D,
                                         N.fp(y), N.fp(scales), N.stream()), "biasnorm_fwd")
        ctx.save_for_backward(x, bias, scales)
        return y

    @staticmethod
    def backward(ctx, g):
        x, bias, scales = ctx.saved_tensors
        D = x.shape[-1]
        g = g.contiguous().float()
        rows = x.numel() // D
        dx = torch.empty_like(x)
        acc = torch.zeros(D + 1, dtype=torch.float32, device=x.device)
        N.PROF[0] and N.profile_note("s2t_biasnorm_bwd", 12.0 * x.numel())
        N.check(N.lib().s2t_biasnorm_bwd(N.fp(x), N.fp(bias), N.fp(scales), N.fp(g), rows, D,
                                         N.fp(dx), N.fp(acc), ctypes_off(acc, D), N.stream()),
                "biasnorm_bwd")
        return dx, acc[:D], acc[D].reshape(())


def ctypes_off(t, n_elems):
    import ctypes
    return ctypes.c_void_p(t.data_ptr() + 4 * n_elems)


def bias_norm(x, bias, log_scale):
    return _BiasNorm.apply(x, bias, log_scale)


class _BiasNormTB(torch.autograd.Function):
    """BiasNorm of a batch-major (B,T,D) tensor whose result is STORED time-major: the returned
    (B,T,D) tensor is a transposed view of a contiguous (T,B,D) buffer, so the encoder's
    `x.transpose(0, 1)` (zipformer.py:183) is contiguous without a copy, and the gradient that comes
    back through it is read in that order."""

    @staticmethod
    def forward(ctx, x, bias, log_scale):
        _dev(x, bias, log_scale)
        Bn, T, D = x.shape
        x = x.contiguous().float()
        y = torch.empty((T, Bn, D), dtype=torch.float32, device=x.device)
        scales = torch.empty(Bn * T, dtype=torch.float32, device=x.device)
        bias = bias.contiguous().float()
        N.PROF[0] and N.profile_note("s2t_biasnorm_fwd_tb", 8.0 * x.numel())
        N.check(N.lib().s2t_biasnorm_fwd_tb(N.fp(x), N.fp(bias), N.fp(log_scale.reshape(1).contiguous().float()),
                                            T, Bn, D, N.fp(y), N.fp(scales), N.stream()), "biasnorm_fwd_tb")
        ctx.save_for_backward(x, bias, scales)
        return y.transpose(0, 1)

    @staticmethod
    def backward(ctx, g):
        x, bias, scales = ctx.saved_tensors
        Bn, T, D = x.shape
        gt = g.transpose(0, 1)                       # (T,B,D): contiguous when it comes from the encoder
        if not gt.is_contiguous() or gt.dtype != torch.float32:
            gt = gt.contiguous().float()
        dx = torch.empty_like(x)
        acc = torch.zeros(D + 1, dtype=torch.float32, device=x.device)
        N.PROF[0] and N.profile_note("s2t_biasnorm_bwd_tb", 12.0 * x.numel())
        N.check(N.lib().s2t_biasnorm_bwd_tb(N.fp(x), N.fp(bias), N.fp(scales), N.fp(gt), T, Bn, D,
                                            N.fp(dx), N.fp(acc), ctypes_off(acc, D), N.stream()),
                "biasnorm_bwd_tb")
        return dx, acc[:D], acc[D].reshape(())


def bias_norm_time_major(x, bias, log_scale):
    """bias_norm(x) for x (B,T,D), stored time-major (see _BiasNormTB)."""
    return _BiasNormTB.apply(x, bias, log_scale)


# ------------------------------------------------------------------ Balancer / Whiten backward
_BAL_WS = {}


def _balancer_workspace(dev):
    """[workspace, call parity]: two alternating statistics accumulators, zeroed once (each call
    clears the one the next call adds into -- zip_elem.hip balancer_apply_fused_kernel)."""
    ws = _BAL_WS.get(dev)
    if ws is None:
        ws = [torch.zeros(N.lib().s2t_balancer_bwd_workspace_floats(), dtype=torch.float32,
                          device=dev), 0]
        _BAL_WS[dev] = ws
    return ws


def balancer_backward(x, g, min_mean, max_mean, min_rms, max_rms, grad_scale, channel_dim,
                      inplace=False, swoosh_l=None):
    """Closed form of reference scaling.py:741-789: the autograd-inside-backward there reduces
    to per-channel statistics (mean, E[x^2]) and a per-element affine term
        g' = g + |g| * grad_scale * (a_c + b_c x) / rms_c(a + b x).
    Channel-last tensors (the zipformer layers) run two HIP launches (s2t_balancer_bwd): column
    statistics + coefficients, fused update.  x and g may be row-strided slices of wider tensors;
    inplace=True writes the result over g (a slice of a gradient being assembled).  swoosh_l
    (True / False): g is the gradient w.r.t. SwooshL / SwooshR of x and goes through the
    activation's derivative first, in the same pass.  Other layouts (NCHW frontend) use the same
    formulas as torch reductions."""
    _dev(x, g)
    nd = x.ndim
    if channel_dim == nd - 1 and x.dim() >= 2 and x.stride(-1) == 1 and g.dtype == torch.float32 \
            and x.dtype == torch.float32 and x.shape[-1] <= 1024:
        C = x.shape[-1]
        x2 = x.reshape(-1, C) if x.is_contiguous() else None
        if x2 is None:
            # row-strided 2-D slice of a wider contiguous tensor (e.g. the gate half of in_proj)
            xs = x.flatten(0, -2) if x.dim() > 2 else x
            if xs.stride(-1) != 1:
                xs = x.contiguous().reshape(-1, C)
            x2 = xs
        rows = x2.shape[0]
        if inplace and g.dim() == 2 and g.stride(1) == 1:
            g2 = out = g
        else:
            g2 = g.contiguous().reshape(-1, C)
            out = torch.empty_like(g2)
        ws = _balancer_workspace(x.device)
        ws[1] = N.lib().s2t_balancer_next_parity()      # (one sequence for this path and zip_layer.hip)
        N.PROF[0] and N.profile_note("s2t_balancer_bwd", 4.0 * rows * C * 4)     # x twice (stats, update), g, out
        N.check(N.lib().s2t_balancer_bwd(N.raw(x2, torch.float32), x2.stride(0),
                                         N.raw(g2, torch.float32), g2.stride(0), rows, C, min_mean,
                                         max_mean, min_rms, max_rms, grad_scale,
                                         N.raw(out, torch.float32), out.stride(0), N.fp(ws[0]),
                                         ws[1], -1.0 if swoosh_l is None else _SW[swoosh_l][0],
                                         N.stream()),
                "s2t_balancer_bwd")
        return out if out is g else out.reshape(g.shape)
    if swoosh_l is not None:
        g = swoosh_backward(x, g, swoosh_l)
    dims = [i for i in range(nd) if i != channel_dim]
    xf = x.float()
    n = xf.numel() // xf.shape[channel_dim]
    mean = xf.mean(dim=dims, keepdim=True)
    uvar = (xf * xf).mean(dim=dims, keepdim=True)
    var = (uvar - mean * mean).clamp(min=1.0e-20)
    std = var.sqrt()
    rms = uvar.clamp(min=1.0e-20).sqrt()
    m = mean / std
    mc = m.clamp(min=min_mean, max=max_mean)
    s_m = torch.sign(m - mc)
    rc = rms.clamp(min=min_rms, max=max_rms)
    s_r = -torch.sign((rc / rms).log())
    # where the clamps on var/uvar were active the reference's autograd sees constants
    live_v = (uvar - mean * mean) > 1.0e-20
    live_r = uvar > 1.0e-20
    inv_n = 1.0 / n
    a = s_m * inv_n * torch.where(live_v, 1.0 / std + mean * mean / (std * var), 1.0 / std)
    b = torch.where(live_v, -s_m * inv_n * mean / (std * var), torch.zeros_like(mean)) + \
        torch.where(live_r, s_r * inv_n / (rms * rms), torch.zeros_like(rms))
    lg_rms = (a * a + 2 * a * b * mean + b * b * uvar).clamp(min=0).sqrt().clamp(min=1.0e-20)
    coef = grad_scale / lg_rms
    gf = g.float()
    return (gf + gf.abs() * ((a * coef) + (b * coef) * xf)).to(g.dtype)


_PINNED = [None, 0]


def _pinned_slot():
    """One float of pinned host memory from a ring (allocated once: hipHostMalloc is slow)."""
    if _PINNED[0] is None:
        _PINNED[0] = torch.zeros(4096, dtype=torch.float32, pin_memory=True)
    i = _PINNED[1]
    _PINNED[1] = (i + 1) % 4096
    return _PINNED[0][i:i + 1]


_WH_SCRATCH = {}


def _whiten_scratch(dev, C):
    """(accumulator (C+1, C), metric-kernel workspace) of the whitening statistics, per channel
    count; both zeroed once -- the metric kernel leaves them clean (whiten.hip)."""
    ent = _WH_SCRATCH.get((dev, C))
    if ent is None:
        ent = (torch.zeros((C + 1, C), dtype=torch.float32, device=dev),
               torch.zeros(4 + 2 * C, dtype=torch.float32, device=dev))
        _WH_SCRATCH[(dev, C)] = ent
    return ent


_STATS_SIDE = os.environ.get("S2T_WHITEN_STREAM", "1") == "1"


def _stats_stream():
    """Side-stream handle for forward-pass statistics, ordered after the work enqueued so far on
    the current stream; None when disabled (then the statistics run on the current stream)."""
    if not (_STATS_SIDE and _Side.enabled):
        return None
    if _Side.handle is None:
        h = N.lib().s2t_side_stream()
        if not h:
            return None
        _Side.handle = ctypes.c_void_p(h)
    N.check(N.lib().s2t_stream_order(N.stream(), _Side.handle), "s2t_stream_order(stats)")
    return _Side.handle


class WhitenStats:
    """Whitening statistics of x, computed when the module fires in FORWARD (they depend on x
    only) so that the scalar `metric` reaches the host through pinned memory long before the
    backward pass needs it: no pipeline-draining read-back inside backward (the reference calls
    .item()-style comparisons there, scaling.py:1012).
    x^T x and the column sums come from one pass of the split-row MFMA kernel; cov = x^T x -
    N mean mean^T per group and the metric from one small kernel (whiten.hip)."""

    def __init__(self, x, num_groups):
        _dev(x)
        C = x.shape[-1]
        xf = x.detach().reshape(-1, C)
        if xf.dtype != torch.float32:
            xf = xf.float()
        n = xf.shape[0]
        G, cg = num_groups, C // num_groups
        dev = x.device
        # x^T x and the column sums in one pass of the TN MFMA GEMM, accumulated into a
        # persistent (C+1, C) buffer that the metric kernel hands back zeroed
        acc, ws = _whiten_scratch(dev, C)
        xtx, colsum = acc[:C], acc[C]
        # The statistics depend on x only and nothing in the forward pass reads them: they run on
        # the library's side stream (ordered after the kernel that produced x), filling the CUs
        # the main chain's small kernels leave idle.  All statistics share one accumulator per
        # channel count; the side stream serialises them in issue order.
        side = _stats_stream() if _tn_ok(xf) else None
        if _tn_ok(xf):
            # symmetric product: only the 64x64 tiles on / above the diagonal that hold same-group
            # pairs are computed (6 of 9 at C = 192, 10 of 16 at 256, the diagonal for the keys)
            N.PROF[0] and N.profile_note("s2t_gemm_xtx", 4.0 * (xf.numel() + C * cg), 2.0 * n * C * cg)
            N.check(N.lib().s2t_gemm_xtx(N.raw(xf, torch.float32), xf.stride(0), n, C, cg,
                                         N.fp(xtx), xtx.stride(0), N.fp(colsum),
                                         side if side is not None else N.stream()), "s2t_gemm_xtx")
        else:
            # layout the TN kernel refuses: library product on the current stream -- which must first
            # be ordered after the side stream, whose earlier statistics use the same accumulator
            if _Side.handle is not None:
                N.check(N.lib().s2t_stream_order(_Side.handle, N.stream()), "s2t_stream_order(stats)")
            a, b = linear_wgrad(xf, xf, True)
            xtx.copy_(a)
            colsum.copy_(b)
        self.cov = torch.empty((G, cg, cg), dtype=torch.float32, device=dev)
        self.mean = torch.empty((C,), dtype=torch.float32, device=dev)
        self.scal = torch.empty((4,), dtype=torch.float32, device=dev)
        self.host = _pinned_slot()
        N.PROF[0] and N.profile_note("s2t_whiten_metric", 4.0 * (2 * C * cg + C * C))
        N.check(N.lib().s2t_whiten_metric(N.fp(xtx), N.fp(colsum), n, G, cg, N.fp(self.cov),
                                          N.fp(self.mean), N.fp(self.scal),
                                          ctypes.c_void_p(self.host.data_ptr()), N.fp(ws),
                                          side if side is not None else N.stream()),
                "s2t_whiten_metric")
        # The round-6 form of the backward (csrc/zip_layer.hip whiten_stats / whiten_bwd,
        # include/s2t_mi355.h s2t_whiten_prep): d metric / d cov, its bias row and dcov's bf16 pieces
        # depend on x only -- taken now, on the statistics' stream
        self.pieces = None
        self.pg = None
        ent = planes.adhoc_entry(C, C, 1, dev) if (_WHITEN_X3P == 2 and X3P["on"] and _tn_ok(xf) and C % 8 == 0
                                                    and cg <= 1024 and n >= 4
                                                    and n * max(C, xf.stride(0)) * 4 < 0x7FFFFF00) else None
        if ent is not None:
            q = side if side is not None else N.stream()
            self.dcov = torch.empty((C, C), dtype=torch.float32, device=dev)
            self.bias = torch.empty((C,), dtype=torch.float32, device=dev)
            self.sums = torch.empty((128,), dtype=torch.float32, device=dev)     # [2][64] partial-sum slots
            self.pieces = torch.empty_like(ent[1])
            N.PROF[0] and N.profile_note("s2t_whiten_prep", 4.0 * (C * cg * cg + C * C))
            N.check(N.lib().s2t_whiten_prep(N.fp(self.cov), N.fp(self.mean), N.fp(self.scal), G, cg,
                                            N.fp(self.dcov), N.fp(self.bias), N.fp(self.sums), q), "s2t_whiten_prep")
            N.PROF[0] and N.profile_note("s2t_x3p_split", 4.0 * C * C + 2.0 * self.pieces.numel())
            N.check(N.lib().s2t_x3p_split(self.dcov.data_ptr(), ent[0].data_ptr(), 1, ent[2],
                                          self.pieces.data_ptr(), q), "s2t_x3p_split")
            if _WHITEN_FWD_PG:
                # the penalty product x dcov + bias depends on x only as well: here, with ||pg||^2 from its
                # epilogue (csrc/zip_layer.hip whiten_stats makes the same calls); backward adds ||g||^2
                self.pg = torch.empty((n, C), dtype=torch.float32, device=dev)
                N.PROF[0] and N.profile_note("s2t_gemm_x3p_sq", 4.0 * (xf.numel() + self.pg.numel()) + 6.0 * C * C,
                                             2.0 * n * C * C)
                two = gemm_arith(_WHITEN_PG2_CLS) == 2
                tile = X3P["tile"] or ((2212 if two else 312) if C % 128 == 0 else (2221 if two else 321))
                with gemm_class(_WHITEN_PG2_CLS):
                    rc = N.lib().s2t_gemm_x3p_sq(N.raw(xf, torch.float32), xf.stride(0),
                                                 ctypes.c_void_p(self.pieces.data_ptr()), C, C, N.fp(self.pg), C, n,
                                                 N.fp(self.bias), None, 0, N.fp(self.sums), tile, q)
                N.check(rc, "s2t_gemm_x3p_sq(forward)")
        if side is not None:
            # the side stream may still be reading x / writing the statistics when this object (or
            # a float() temporary of x) dies -- e.g. a forward under train() with no backward: the
            # caching allocator must not hand the memory to main-stream work before the join
            _Side.keep.append((xf, self.cov, self.mean, self.scal) +
                              ((self.dcov, self.bias, self.sums, self.pieces, self.pg) if self.pieces is not None else ()))
        self.event = torch.cuda.Event()
        if side is not None:
            self.event.record(N._launch_stream((side,)))
        else:
            self.event.record()
        self.num_groups, self.cg = G, cg

    def metric(self):
        self.event.synchronize()
        return float(self.host[0])


# (round 5: off -- the NN kernel that takes the two norms in its epilogue (below) beats the bf16x3
# product + a separate norm pass also for the 31 680-row activations: 37.61 against 37.72 ms/step)
# (round 6: 2 = dcov and its pieces taken in forward on the statistics' stream, backward = the penalty
# product on the pre-split-weight kernel with the two norms in its epilogue + the combining pass; 0 = the
# three-launch form on the NN kernel -- also what shapes outside the pre-split kernel's rules take)
_WHITEN_X3P = int(os.environ.get("S2T_WHITEN_X3P", "2"))
# the norms of (g, x dcov) taken in the product's epilogue (s2t_gemm_f32_sq) instead of by a pass over both
_WHITEN_SQ = os.environ.get("S2T_WHITEN_SQ", "1") == "1"
# the penalty product x dcov itself in forward, on the statistics' stream (late round 6; parity-tested).  OFF:
# measured 33.49 / 33.46 / 33.12 against 32.78 / 32.79 / 32.63 ms per step with the product in backward -- a
# GEMM with a 48-64 KB / 130-200 register footprint on the side stream costs forward's own GEMMs more than
# the 25 us per firing Whiten it takes off backward's chain
_WHITEN_FWD_PG = os.environ.get("S2T_WHITEN_FWD_PG", "0") == "1"


_WHITEN_PG_CLS = int(os.environ.get("S2T_WHITEN_PG_CLS", "3"))   # class of the penalty product, three-launch form (csrc/zip_layer.hip whiten_bwd)
_WHITEN_PG2_CLS = int(os.environ.get("S2T_WHITEN_PG_CLS", "1"))  # ... on the pre-split-weight kernel (round-6 form): data gradient


def whiten_backward(x, g, stats, limit, grad_scale):
    """Closed form of reference scaling.py:949-1028.  Returns (grad, penalty_was_active).
    d metric/d x = 2 (x - mean) dcov  (the centring's own Jacobian vanishes because the centred
    columns sum to zero), i.e. ONE GEMM with a bias row; the norm ratio and the final axpy are
    two more launches."""
    if not (stats.metric() >= limit):
        return g, False
    shp = x.shape
    C = shp[-1]
    G, cg = stats.num_groups, stats.cg
    dev = x.device
    if getattr(stats, "pieces", None) is not None:
        xf = x.reshape(-1, C)
        g2 = g.contiguous().float()
        if stats.pg is not None and g2.data_ptr() % 16 == 0:
            # the product and ||pg||^2 were taken in forward (stats.metric() waited for their stream)
            out = torch.empty_like(g2)
            N.PROF[0] and N.profile_note("s2t_sumsq64", 4.0 * g2.numel())
            N.check(N.lib().s2t_sumsq64(N.fp(g2), g2.numel(), N.fp(stats.sums), N.stream()), "s2t_sumsq64")
            N.PROF[0] and N.profile_note("s2t_whiten_combine64", 12.0 * g2.numel())
            N.check(N.lib().s2t_whiten_combine64(N.fp(g2), N.fp(stats.pg), g2.numel(), float(grad_scale),
                                                 N.fp(stats.sums), N.fp(out), N.stream()), "s2t_whiten_combine64")
            return out.view(shp), True
        if xf.dtype is torch.float32 and xf.stride(1) == 1 and g2.data_ptr() % 16 == 0:
            out = torch.empty_like(g2)
            pg = torch.empty_like(g2)
            N.PROF[0] and N.profile_note("s2t_gemm_x3p_sq", 4.0 * (xf.numel() + 2 * g2.numel()) + 6.0 * C * C,
                                         2.0 * xf.shape[0] * C * C)
            # (class and block tile as csrc/zip_layer.hip whiten_bwd: a data-gradient-like product, two
            #  pieces by default; 128-wide column tiles where C is a multiple of 128, 64-wide otherwise)
            two = gemm_arith(_WHITEN_PG2_CLS) == 2
            tile = X3P["tile"] or ((2212 if two else 312) if C % 128 == 0 else (2221 if two else 321))
            with gemm_class(_WHITEN_PG2_CLS):
                rc = N.lib().s2t_gemm_x3p_sq(N.raw(xf, torch.float32), xf.stride(0),
                                             ctypes.c_void_p(stats.pieces.data_ptr()), C, C, N.fp(pg), C,
                                             xf.shape[0], N.fp(stats.bias), N.fp(g2), C, N.fp(stats.sums),
                                             tile, N.stream())
            N.check(rc, "s2t_gemm_x3p_sq")
            N.PROF[0] and N.profile_note("s2t_whiten_combine64", 12.0 * g2.numel())
            N.check(N.lib().s2t_whiten_combine64(N.fp(g2), N.fp(pg), g2.numel(), float(grad_scale),
                                                 N.fp(stats.sums), N.fp(out), N.stream()), "s2t_whiten_combine64")
            return out.view(shp), True
    dcov = torch.empty((C, C), dtype=torch.float32, device=dev)
    bias = torch.empty((C,), dtype=torch.float32, device=dev)
    sums = torch.empty((2,), dtype=torch.float32, device=dev)
    N.PROF[0] and N.profile_note("s2t_whiten_dcov", 4.0 * (C * cg + C * C + 2 * C))
    N.check(N.lib().s2t_whiten_dcov(N.fp(stats.cov), N.fp(stats.mean), N.fp(stats.scal), G, cg,
                                    N.fp(dcov), N.fp(bias), N.fp(sums), N.stream()),
            "s2t_whiten_dcov")
    xf = x.reshape(-1, C)
    if xf.dtype != torch.float32:
        xf = xf.float()
    pg = None
    g2 = g.contiguous().float()
    if g2.data_ptr() % 16:
        g2 = g2.clone()
    out = torch.empty_like(g2)
    with gemm_class(_WHITEN_PG_CLS):               # (the penalty's product x dcov: a statistic)
        return _whiten_penalty(xf, g2, dcov, bias, sums, pg, out, C, shp, grad_scale, dev)


def _whiten_penalty(xf, g2, dcov, bias, sums, pg, out, C, shp, grad_scale, dev):
    if pg is None:
        # (our NN kernel with the bias in its epilogue: the same calls csrc/zip_layer.hip makes);
        # first the form that takes the two norms of (g, pg) while pg leaves the accumulators
        pg = torch.empty((xf.shape[0], C), dtype=torch.float32, device=dev)
        if _WHITEN_SQ and xf.stride(1) == 1:
            N.PROF[0] and N.profile_note("s2t_gemm_f32_sq", 4.0 * (xf.numel() + 2 * pg.numel() + C * C),
                                         2.0 * xf.shape[0] * C * C)
            rc = N.lib().s2t_gemm_f32_sq(1, N.raw(xf, torch.float32), xf.stride(0), N.fp(dcov), C, N.fp(pg), C,
                                         xf.shape[0], C, C, N.fp(bias), N.fp(g2.view(-1, C)), C, N.fp(sums),
                                         N.stream())
            if rc == 0:
                N.PROF[0] and N.profile_note("s2t_whiten_combine", 12.0 * g2.numel())
                N.check(N.lib().s2t_whiten_combine(N.fp(g2), N.fp(pg), g2.numel(), float(grad_scale),
                                                   N.fp(sums), N.fp(out), N.stream()), "s2t_whiten_combine")
                return out.view(shp), True
            if rc != -2:
                N.check(rc, "s2t_gemm_f32_sq(whiten)")
        N.PROF[0] and N.profile_note("s2t_gemm_f32", 4.0 * (xf.numel() + pg.numel() + C * C), 2.0 * xf.shape[0] * C * C)
        rc = N.lib().s2t_gemm_f32(1, N.raw(xf, torch.float32), xf.stride(0), N.fp(dcov), C, N.fp(pg), C,
                                  xf.shape[0], C, C, N.fp(bias), None, 0, None, 0, 0, 0, 0, None, 0,
                                  N.stream()) if xf.stride(1) == 1 else -2
        if rc == -2:
            pg = torch.addmm(bias, xf, dcov)
        else:
            N.check(rc, "s2t_gemm_f32(whiten)")
    N.PROF[0] and N.profile_note("s2t_whiten_apply", 12.0 * g2.numel())
    N.check(N.lib().s2t_whiten_apply(N.fp(g2), N.fp(pg), g2.numel(), float(grad_scale), N.fp(sums),
                                     N.fp(out), N.stream()), "s2t_whiten_apply")
    return out.view(shp), True


def limit_param_grad(x, g, lo, hi):
    _dev(x, g)
    xc, gc = x.contiguous().float(), g.contiguous().float()
    out = torch.empty_like(gc)
    N.PROF[0] and N.profile_note("s2t_limit_param_grad", 12.0 * gc.numel())
    N.check(N.lib().s2t_limit_param_grad(N.fp(xc), N.fp(gc), float(lo), float(hi), gc.numel(),
                                         N.fp(out), N.stream()), "s2t_limit_param_grad")
    return out.view(g.shape)


# ------------------------------------------------------------------ conv module core
def conv_params(conv, T, chunk_size):
    """-> (chunk, K, wc, bc, wk, bk, scale): the parameters of a ChunkCausalDepthwiseConv1d-like
    module (causal_conv / chunkwise_conv / chunkwise_conv_scale) or of a depthwise nn.Conv1d
    (wc, bc, scale = None) and the chunk length the kernel works with."""
    if isinstance(conv, torch.nn.Conv1d):
        K = conv.kernel_size[0]
        assert conv.groups == conv.in_channels and conv.padding[0] == K // 2
        return max(T, 1), K, None, None, conv.weight, conv.bias, None
    chunk = T if (chunk_size < 0 or chunk_size > T) else chunk_size
    return (chunk, conv.kernel_size, conv.causal_conv.weight, conv.causal_conv.bias,
            conv.chunkwise_conv.weight, conv.chunkwise_conv.bias, conv.chunkwise_conv_scale)


def zipconv_forward(u, gate_off, m8, chunk, K, wc, bc, wk, bk, scale, act=None):
    """u (T,B,ld) contiguous fp32 -> y (T,B,C); with act = True (SwooshL) / False (SwooshR) ->
    (y, act(y)), the activation written by the same pass.   HIP: zip_conv.hip."""
    T, B, ld = u.shape
    C = wk.shape[0]
    y = torch.empty((T, B, C), dtype=torch.float32, device=u.device)
    if act is not None:
        ya = torch.empty_like(y)
        N.PROF[0] and N.profile_note("s2t_zipconv_fwd_act", 4.0 * (u.numel() + 2 * y.numel()))
        N.check(N.lib().s2t_zipconv_fwd_act(N.fp(u), ld, gate_off, N.ptr(m8), T, B, C, K, chunk,
                                            N.fp(wc), N.fp(bc), N.fp(wk), N.fp(bk), N.fp(scale), N.fp(y),
                                            N.fp(ya), 1 if act else 2, N.stream()), "s2t_zipconv_fwd_act")
        return y, ya
    N.PROF[0] and N.profile_note("s2t_zipconv_fwd", 4.0 * (u.numel() + y.numel()))
    N.check(N.lib().s2t_zipconv_fwd(N.fp(u), ld, gate_off, N.ptr(m8), T, B, C, K, chunk,
                                    N.fp(wc), N.fp(bc), N.fp(wk), N.fp(bk), N.fp(scale), N.fp(y),
                                    N.stream()), "s2t_zipconv_fwd")
    return y


_CONV_W_SIDE = os.environ.get("S2T_CONV_W_SIDE", "1") == "1"


def zipconv_backward(u, gate_off, m8, chunk, K, wc, wk, bk, scale, dy, grads, side=True):
    """-> du (T,B,2C | C).  grads = (dwc, dbc, dwk, dbk, dscale) tensors the kernels ACCUMULATE the
    parameter gradients into (None where the parameter is absent).  side: the parameter-gradient
    kernel may run on the side stream -- ONLY when `grads` are the flat-store views, which nothing
    reads before the end-of-backward join; a scratch buffer that is handed back to autograd
    (AccumulateGrad reads it on the main stream right away, and the allocator may reuse it) must
    be written on the current stream."""
    T, B, ld = u.shape
    C = wk.shape[0]
    dev = u.device
    du = torch.empty((T, B, 2 * C if gate_off >= 0 else C), dtype=torch.float32, device=dev)
    ws = torch.empty(N.lib().s2t_zipconv_bwd_workspace_floats(T, B, C, K), dtype=torch.float32,
                     device=dev)
    dwc, dbc, dwk, dbk, dsc = grads
    # the tap / bias / edge-scale gradients only feed the optimizer: side stream, as the weight-
    # gradient GEMMs (operands kept alive until the join)
    wst = _side_launch_stream(u, dy, ws, m8, wc, wk, bk, scale) if (_CONV_W_SIDE and side) else None
    L = N.lib()
    gptr = (N.raw(dwc) if dwc is not None else None, N.raw(dbc) if dbc is not None else None,
            N.raw(dwk), N.raw(dbk) if dbk is not None else None,
            N.raw(dsc) if dsc is not None else None)
    if wst is None:
        N.PROF[0] and N.profile_note("s2t_zipconv_bwd", 4.0 * (2 * u.numel() + 2 * dy.numel()))
        N.check(L.s2t_zipconv_bwd(N.fp(u), ld, gate_off, N.ptr(m8), T, B, C, K, chunk, N.fp(wc),
                                  N.fp(wk), N.fp(bk), N.fp(scale), N.fp(dy), N.fp(du), *gptr,
                                  N.fp(ws), N.stream()), "s2t_zipconv_bwd")
        return du
    N.PROF[0] and N.profile_note("s2t_zipconv_bwd_data", 4.0 * (u.numel() + dy.numel() + du.numel()))
    N.check(L.s2t_zipconv_bwd_data(N.fp(u), ld, gate_off, N.ptr(m8), T, B, C, K, chunk, N.fp(wc),
                                   N.fp(wk), N.fp(bk), N.fp(scale), N.fp(dy), N.fp(du), N.stream()),
            "s2t_zipconv_bwd_data")
    N.PROF[0] and N.profile_note("s2t_zipconv_bwd_params", 4.0 * (u.numel() + dy.numel()))
    N.check(L.s2t_zipconv_bwd_params(N.fp(u), ld, gate_off, N.ptr(m8), T, B, C, K, chunk, N.fp(wc),
                                     N.fp(wk), N.fp(bk), N.fp(scale), N.fp(dy), *gptr, N.fp(ws), wst),
            "s2t_zipconv_bwd_params")
    return du


def direct_grads(params):
    """The flat-store gradient views of `params` (None entries stay None) when every one of them
    is a leaf living in a FlatStore, else None."""
    out = []
    for p in params:
        if p is None:
            out.append(None)
            continue
        if not (p.is_leaf and flat.owned(p)):
            return None
        g = p.grad
        if g is None or not g.is_contiguous():
            return None
        out.append(g)
    return out


class _ZipConv(torch.autograd.Function):
    """Fused gate + padding mask + (chunk-causal | plain) depthwise conv, time-major."""

    @staticmethod
    def forward(ctx, u, gate_off, mask, chunk, K, wc, bc, wk, bk, scale):
        _dev(u, wk)
        u = u.contiguous().float()
        m8 = None if mask is None else mask.to(torch.uint8).contiguous()
        cont = [None if t is None else t.contiguous() for t in (wc, bc, wk, bk, scale)]
        y = zipconv_forward(u, gate_off, m8, chunk, K, *cont)
        ctx.save_for_backward(u, m8, cont[0], cont[2], cont[3], cont[4])
        ctx.cfg = (gate_off, chunk, K, bc is not None)
        ctx.params = (wc, bc, wk, bk, scale)
        return y

    @staticmethod
    def backward(ctx, dy):
        u, m8, wc, wk, bk, scale = ctx.saved_tensors
        gate_off, chunk, K, has_bc = ctx.cfg
        dy = dy.contiguous().float()
        C = wk.shape[0]
        Kh = (K + 1) // 2
        # parameter gradients are ACCUMULATED by the kernels (reduce pass / atomics): when every
        # parameter lives in the flat store they go straight into its gradient views, otherwise
        # into one zeroed scratch buffer that is handed back to autograd
        grads = direct_grads(ctx.params)
        if grads is not None:
            du = zipconv_backward(u, gate_off, m8, chunk, K, wc, wk, bk, scale, dy, grads)
            return (du,) + (None,) * 9
        sizes = [C * Kh if wc is not None else 0, C if wc is not None else 0, C * K,
                 C if bk is not None else 0, 2 * C * K if scale is not None else 0]
        buf = torch.zeros(sum(sizes), dtype=torch.float32, device=u.device)
        parts, o = [], 0
        for n in sizes:
            parts.append(buf[o:o + n] if n else None)
            o += n
        dwc, dbc, dwk, dbk, dsc = parts
        du = zipconv_backward(u, gate_off, m8, chunk, K, wc, wk, bk, scale, dy, parts, side=False)
        return (du, None, None, None, None,
                None if dwc is None else dwc.view(C, 1, Kh),
                dbc if has_bc else None, dwk.view(C, 1, K), dbk,
                None if dsc is None else dsc.view(2, C, K))


def glu_chunk_causal_dwconv(u, gate_off, key_padding_mask, conv, chunk_size):
    """u (T,B,ld): x = u[..., :C], gate pre-activation = u[..., gate_off:gate_off+C]
    (gate_off None: no gate) -> y (T,B,C) = dwconv((x * sigmoid(gate)) zeroed on padded frames).
    conv: ChunkCausalDepthwiseConv1d-like module or a depthwise nn.Conv1d.   HIP: zip_conv.hip."""
    go = -1 if gate_off is None else int(gate_off)
    chunk, K, wc, bc, wk, bk, scale = conv_params(conv, u.shape[0], chunk_size)
    return _ZipConv.apply(u, go, key_padding_mask, chunk, K, wc, bc, wk, bk, scale)


# ------------------------------------------------------------------ attention
class AttnShared:
    """Per layer-call scratch that lets the attention-weights backward contract its consumers'
    gradients on the fly instead of receiving three materialised (H,B,T,T) tensors:
    `pairs` = [(dO, v, O, dv)] from the value-apply consumers, `dW0` = (B,T,T) gradient of the
    head-0 slice (nonlinear attention), `real` = any ordinary (materialised) gradient."""

    def __init__(self):
        self.pairs = []
        self.dW0 = None
        self.real = None


class DeferredWeights:
    """An alias of W handed to one consumer, plus the shared scratch."""

    def __init__(self, w, shared):
        self.w, self.shared = w, shared

    @property
    def shape(self):
        return self.w.shape


def _attn_bwd_call(qkp, pos, k8, a8, H, qd, pd, W, dW, dW0, pairs, delta):
    T, B, _ = qkp.shape
    dev = qkp.device
    dqkp = torch.empty_like(qkp)
    dpos = None if pos is None else torch.empty_like(pos)      # cleared by the entry point
    given = delta is not None
    if delta is None:
        delta = torch.empty((H, B, T), dtype=torch.float32, device=dev)
    p = list(pairs) + [(None, None, None, 0)] * (2 - len(pairs))
    ws = None
    if pos is not None:
        ws = torch.empty(N.lib().s2t_relpos_attn_bwd_workspace_floats(T, B, H, pd),
                         dtype=torch.float32, device=dev)
    # flops per score element: dW from its factors (2 cd), dq and dk (2 qd each), dp and dpos
    # (2 pd each) -- 192 at the C3 dims against 8 bytes: above the chip's 19.7 flop/byte balance,
    # so the f32 matrix cores bound this launch, not HBM
    cd = sum(int(e[3]) for e in p)
    N.PROF[0] and N.profile_note("s2t_relpos_attn_bwd", 4.0 * (2 * qkp.numel() + 2 * W.numel()),
                   2.0 * W.numel() * (cd + 2 * qd + (2 * pd if pos is not None else 0)))
    N.check(N.lib().s2t_relpos_attn_bwd(N.fp(qkp), N.fp(pos), N.ptr(k8), N.ptr(a8), T, B, H, qd, pd,
                                        N.fp(W), N.fp(dW), N.fp(dW0), N.fp(p[0][0]), N.fp(p[0][1]),
                                        p[0][3], N.fp(p[1][0]), N.fp(p[1][1]), p[1][3], int(given),
                                        N.fp(delta), N.fp(dqkp), N.fp(dpos), N.fp(ws), N.stream()),
            "s2t_relpos_attn_bwd")
    return dqkp, dpos


class _RelPosAttn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkp, pos_proj, kpm, amask, H, qd, pd, shared):
        _dev(qkp, pos_proj)
        qkp = qkp.contiguous().float()
        T, B, _ = qkp.shape
        pos = None if pos_proj is None else pos_proj.contiguous().float()
        k8 = None if kpm is None else kpm.to(torch.uint8).contiguous()
        a8 = None if amask is None else amask.to(torch.uint8).contiguous()
        W = torch.empty((H, B, T, T), dtype=torch.float32, device=qkp.device)
        N.PROF[0] and N.profile_note("s2t_relpos_attn_fwd", 4.0 * (qkp.numel() + W.numel()),
                       2.0 * W.numel() * (qd + (pd if pos is not None else 0)))
        N.check(N.lib().s2t_relpos_attn_fwd(N.fp(qkp), N.fp(pos), N.ptr(k8), N.ptr(a8), T, B, H,
                                            qd, pd, N.fp(W), N.stream()), "s2t_relpos_attn_fwd")
        ctx.save_for_backward(qkp, pos, k8, a8, W)
        ctx.cfg = (H, qd, pd)
        ctx.shared = shared
        return W

    @staticmethod
    def backward(ctx, dW):
        qkp, pos, k8, a8, W = ctx.saved_tensors
        H, qd, pd = ctx.cfg
        sh = ctx.shared
        if sh is None:
            dqkp, dpos = _attn_bwd_call(qkp, pos, k8, a8, H, qd, pd, W, dW.contiguous().float(),
                                        None, [], None)
            return dqkp, dpos, None, None, None, None, None, None
        # deferred: the incoming gradient is the fan-out's stride-0 placeholder (ignored) unless
        # W was consumed directly by ordinary ops; contract the stashed factors
        T, B, _ = qkp.shape
        if dW is not None and any(st != 0 for st in dW.stride()):
            sh.real = dW if sh.real is None else sh.real + dW
        real = None if sh.real is None else sh.real.contiguous().float()
        delta = torch.zeros((H, B, T), dtype=torch.float32, device=qkp.device)
        for dO, v, O, dv in sh.pairs:
            delta += (dO * O).view(T, B, H, dv).sum(dim=-1).permute(2, 1, 0)
        if sh.dW0 is not None:
            delta[0] += (W[0] * sh.dW0).sum(dim=-1)
        if real is not None:
            delta += (W * real).sum(dim=-1)
        dqkp, dpos = _attn_bwd_call(qkp, pos, k8, a8, H, qd, pd, W, real, sh.dW0, sh.pairs, delta)
        sh.pairs, sh.dW0, sh.real = [], None, None
        return dqkp, dpos, None, None, None, None, None, None


class _AttnFanout(torch.autograd.Function):
    """W -> n aliases with ONE autograd edge back into the weights node.  Consumers that defer
    their gradient return None; anything else is summed into shared.real."""

    @staticmethod
    def forward(ctx, W, shared, n):
        ctx.shared = shared
        ctx.shape = W.shape
        ctx.set_materialize_grads(False)
        return tuple(W.view_as(W) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        real = None
        for g in grads:
            if g is not None:
                real = g if real is None else real + g
        ctx.shared.real = real
        ref = grads[0] if grads[0] is not None else None
        dev = ref.device if ref is not None else ctx.shared.device
        return torch.zeros((), dtype=torch.float32, device=dev).expand(ctx.shape), None, None


def attn_fanout(W, n=3):
    shared = AttnShared()
    shared.device = W.device
    aliases = _AttnFanout.apply(W, shared, n)
    return [DeferredWeights(a, shared) for a in aliases], shared


class _AttnHead0(torch.autograd.Function):
    @staticmethod
    def forward(ctx, W, shared):
        ctx.shared = shared
        return W[0:1]

    @staticmethod
    def backward(ctx, g):
        g0 = g[0].contiguous().float()
        sh = ctx.shared
        sh.dW0 = g0 if sh.dW0 is None else sh.dW0 + g0
        return None, None


def attn_head0(dw: DeferredWeights):
    """W[0:1] for the nonlinear-attention module; its gradient is stashed as dW0."""
    return _AttnHead0.apply(dw.w, dw.shared)


class _AttnApplyDeferred(torch.autograd.Function):
    @staticmethod
    def forward(ctx, W, v, shared, H):
        _dev(W, v)
        v = v.contiguous().float()
        T, B, HD = v.shape
        dv = HD // H
        out = torch.empty_like(v)
        N.PROF[0] and N.profile_note("s2t_attn_apply", 4.0 * (W.numel() + 2 * v.numel()))
        N.check(N.lib().s2t_attn_apply(N.fp(W), N.fp(v), T, B, H, dv, 0, N.fp(out), N.stream()),
                "s2t_attn_apply")
        ctx.save_for_backward(W, v, out)
        ctx.shared, ctx.H = shared, H
        return out

    @staticmethod
    def backward(ctx, dO):
        W, v, out = ctx.saved_tensors
        H = ctx.H
        T, B, HD = v.shape
        dv = HD // H
        dO = dO.contiguous().float()
        dV = torch.empty_like(v)
        N.check(N.lib().s2t_attn_apply(N.fp(W), N.fp(dO), T, B, H, dv, 1, N.fp(dV), N.stream()),
                "s2t_attn_apply(T)")
        sh = ctx.shared
        if len(sh.pairs) < 2 and sum(p[3] for p in sh.pairs) + dv <= 32:
            sh.pairs.append((dO, v, out, dv))
            return None, dV, None, None
        # no room to defer: materialise this consumer's dW = dO . v^T
        g = torch.matmul(dO.view(T, B, H, dv).permute(2, 1, 0, 3),
                         v.view(T, B, H, dv).permute(2, 1, 3, 0))
        return g, dV, None, None


def relpos_attention_weights(qkp, pos_proj, num_heads, query_head_dim, pos_head_dim, attn_mask,
                             key_padding_mask, penalize=None, shared=None):
    """qkp (T,B,H*(2*qd+pd)) = in_proj(x); pos_proj (2T-1, H*pd) = linear_pos(pos_emb) or None
    -> softmax weights (H,B,T,T).  scores[h,b,i,j] = q_i.k_j + p_i.pos[(T-1)-i+j]; masked
    entries are set to -1000 (reference zipformer.py:1966-2066).   HIP: zip_attn.hip.
    `penalize` (the reference's 10%-of-calls penalize_abs_values_gt on the raw scores, a
    backward-only term) needs the raw scores as an autograd node, so that branch keeps the
    materialised torch composition."""
    if penalize is None:
        return _RelPosAttn.apply(qkp, pos_proj, key_padding_mask, attn_mask, num_heads,
                                 query_head_dim, pos_head_dim, shared)
    T, B, _ = qkp.shape
    H, qd, pd = num_heads, query_head_dim, pos_head_dim
    q = qkp[..., :H * qd].reshape(T, B, H, qd).permute(2, 1, 0, 3)
    k = qkp[..., H * qd:2 * H * qd].reshape(T, B, H, qd).permute(2, 1, 3, 0)
    p = qkp[..., 2 * H * qd:].reshape(T, B, H, pd).permute(2, 1, 0, 3)
    scores = torch.matmul(q, k)
    if pos_proj is not None:
        pe = pos_proj.reshape(1, 2 * T - 1, H, pd).permute(2, 0, 3, 1)
        ps = torch.matmul(p, pe)
        ps = ps.as_strided((H, B, T, T), (ps.stride(0), ps.stride(1), ps.stride(2) - ps.stride(3),
                                          ps.stride(3)), storage_offset=ps.stride(3) * (T - 1))
        scores = scores + ps
    scores = penalize(scores)
    if attn_mask is not None:
        scores = scores.masked_fill(attn_mask, -1000)
    if key_padding_mask is not None:
        scores = scores.masked_fill(key_padding_mask.unsqueeze(1), -1000)
    return scores.softmax(dim=-1)


def attention_apply(weights, v, num_heads):
    """weights (H,B,T,T) [Tensor or DeferredWeights], v (T,B,H*dv) -> (T,B,H*dv).
    DeferredWeights + dv <= 16: HIP apply kernel, gradient w.r.t. W deferred to the weights'
    backward (never materialised).  Otherwise a plain batched GEMM (rocBLAS)."""
    if isinstance(weights, DeferredWeights):
        if v.shape[-1] // num_heads <= 16:
            return _AttnApplyDeferred.apply(weights.w, v, weights.shared, num_heads)
        weights = weights.w
    T, B, _ = v.shape
    x = v.reshape(T, B, num_heads, -1).permute(2, 1, 0, 3)
    x = torch.matmul(weights, x)
    return x.permute(2, 1, 0, 3).reshape(T, B, -1)


# ------------------------------------------------------------------ misc streaming ops
class _Bypass(torch.autograd.Function):
    """orig + (src - orig) * scale[c] in one pass; backward = one pass + per-channel atomics."""

    @staticmethod
    def forward(ctx, orig, src, scale):
        orig, src = orig.contiguous().float(), src.contiguous().float()
        scale = scale.contiguous().float()
        C = src.shape[-1]
        out = torch.empty_like(src)
        N.PROF[0] and N.profile_note("s2t_bypass_fwd", 12.0 * src.numel())
        N.check(N.lib().s2t_bypass_fwd(N.fp(orig), N.fp(src), N.fp(scale), src.numel() // C, C,
                                       N.fp(out), N.stream()), "s2t_bypass_fwd")
        ctx.save_for_backward(orig, src, scale)
        return out

    @staticmethod
    def backward(ctx, g):
        orig, src, scale = ctx.saved_tensors
        g = g.contiguous().float()
        C = src.shape[-1]
        d_orig, d_src = torch.empty_like(src), torch.empty_like(src)
        d_scale = torch.zeros_like(scale)
        N.PROF[0] and N.profile_note("s2t_bypass_bwd", 20.0 * src.numel())
        N.check(N.lib().s2t_bypass_bwd(N.fp(orig), N.fp(src), N.fp(scale), N.fp(g),
                                       src.numel() // C, C, N.fp(d_orig), N.fp(d_src),
                                       N.fp(d_scale), N.stream()), "s2t_bypass_bwd")
        return d_orig, d_src, d_scale


def bypass_combine(src_orig, src, scale):
    """BypassModule core (reference zipformer.py:1523-1555).  HIP: zip_glue.hip."""
    if scale.dim() == 1 and src.is_cuda and src.shape[-1] % 4 == 0 and src.shape == src_orig.shape \
            and src.data_ptr() % 16 == 0 and src_orig.data_ptr() % 16 == 0:
        return _Bypass.apply(src_orig, src, scale)
    return src_orig + (src - src_orig) * scale          # per-utterance skip / straight-through masks


class _NonlinCore(torch.autograd.Function):
    """NonlinAttention between in_proj and out_proj (reference zipformer.py:2459-2478):
    u = [s | x | y] -> (W0 @ (x * tanh(s))) * y, with the Balancer on s and the Whiten on x (both
    identity in forward) applied to the slices of the gradient inside this backward.  Three
    launches forward (gate -> batch-major, rocBLAS bmm, out), five backward; no chunk / cat /
    permute copies."""

    @staticmethod
    def forward(ctx, u, w0, bal_cfg, whiten_mod):
        _dev(u, w0)
        u = u.contiguous().float()
        T, B, C3 = u.shape
        C = C3 // 3
        L = N.lib()
        st = N.stream()
        xs = torch.empty((B, T, C), dtype=torch.float32, device=u.device)
        N.PROF[0] and N.profile_note("s2t_nonlin_gate_fwd", 12.0 * T * B * C)
        N.check(L.s2t_nonlin_gate_fwd(N.fp(u), T, B, C, N.fp(xs), st), "nonlin_gate_fwd")
        wm = w0.reshape(B, T, T)
        z = torch.bmm(wm, xs)                                         # rocBLAS
        o = torch.empty((T, B, C), dtype=torch.float32, device=u.device)
        N.PROF[0] and N.profile_note("s2t_nonlin_out_fwd", 12.0 * T * B * C)
        N.check(L.s2t_nonlin_out_fwd(N.fp(z), N.fp(u), T, B, C, N.fp(o), st), "nonlin_out_fwd")
        ctx.save_for_backward(u, wm, xs, z)
        ctx.bal_cfg, ctx.whiten_mod = bal_cfg, whiten_mod
        ctx.wshape = w0.shape
        ctx.stats = None
        if whiten_mod is not None:
            ctx.stats = WhitenStats(u[..., C:2 * C], whiten_mod.num_groups)
        return o

    @staticmethod
    def backward(ctx, g):
        u, wm, xs, z = ctx.saved_tensors
        T, B, C3 = u.shape
        C = C3 // 3
        L = N.lib()
        st = N.stream()
        g = g.contiguous().float()
        dz = torch.empty_like(z)
        du = torch.empty_like(u)
        N.PROF[0] and N.profile_note("s2t_nonlin_out_bwd", 20.0 * T * B * C)
        N.check(L.s2t_nonlin_out_bwd(N.fp(g), N.fp(z), N.fp(u), T, B, C, N.fp(dz), N.fp(du), st),
                "nonlin_out_bwd")
        dxs = torch.bmm(wm.transpose(1, 2), dz)
        dW0 = torch.bmm(dz, xs.transpose(1, 2)) if ctx.needs_input_grad[1] else None
        N.PROF[0] and N.profile_note("s2t_nonlin_gate_bwd", 20.0 * T * B * C)
        N.check(L.s2t_nonlin_gate_bwd(N.fp(dxs), N.fp(u), T, B, C, N.fp(du), st), "nonlin_gate_bwd")
        if ctx.bal_cfg is not None:
            du[..., :C] = balancer_backward(u[..., :C], du[..., :C].contiguous(), *ctx.bal_cfg[:5],
                                            2)
        if ctx.whiten_mod is not None:
            wmod = ctx.whiten_mod
            out, active = whiten_backward(u[..., C:2 * C], du[..., C:2 * C].contiguous(), ctx.stats,
                                          float(wmod.whitening_limit), float(wmod.grad_scale))
            if active:
                du[..., C:2 * C] = out
            wmod.prob = wmod.max_prob if active else wmod.min_prob
        return du, (None if dW0 is None else dW0.view(ctx.wshape)), None, None


def nonlin_core(u, w0, bal_cfg=None, whiten_mod=None):
    return _NonlinCore.apply(u, w0, bal_cfg, whiten_mod)


class _Downsample(torch.autograd.Function):
    """sum_k w[k] * src[min(tt*ds + k, T-1)]: one pass forward, one pass backward (d_src and the
    tap gradients; the softmax over the learnable bias stays in autograd)."""

    @staticmethod
    def forward(ctx, src, w, ds, batch_major=False):
        src = src.contiguous().float()
        w = w.contiguous().float()
        T, B, C = src.shape
        dT = (T + ds - 1) // ds
        N.PROF[0] and N.profile_note("s2t_downsample_fwd_bt" if batch_major else "s2t_downsample_fwd",
                                     4.0 * (src.numel() + dT * B * C))
        ctx.save_for_backward(src, w)
        ctx.ds, ctx.bm = ds, bool(batch_major)
        if batch_major:
            # stored (B, dT, C): the returned (dT, B, C) tensor is a transposed view, so the caller's
            # x.transpose(0, 1) is contiguous without a copy (and so is the gradient coming back)
            out = torch.empty((B, dT, C), dtype=torch.float32, device=src.device)
            N.check(N.lib().s2t_downsample_fwd_bt(N.fp(src), N.fp(w), ds, T, B, C, N.fp(out), N.stream()),
                    "s2t_downsample_fwd_bt")
            return out.transpose(0, 1)
        out = torch.empty((dT, B, C), dtype=torch.float32, device=src.device)
        N.check(N.lib().s2t_downsample_fwd(N.fp(src), N.fp(w), ds, T, B, C, N.fp(out), N.stream()),
                "s2t_downsample_fwd")
        return out

    @staticmethod
    def backward(ctx, g):
        src, w = ctx.saved_tensors
        T, B, C = src.shape
        d_src = torch.empty_like(src)
        dw = torch.zeros_like(w)
        N.PROF[0] and N.profile_note("s2t_downsample_bwd_bt" if ctx.bm else "s2t_downsample_bwd",
                                     4.0 * (2 * src.numel() + g.numel()))
        if ctx.bm:
            gb = g.transpose(0, 1)
            if not gb.is_contiguous() or gb.dtype != torch.float32:
                gb = gb.contiguous().float()
            N.check(N.lib().s2t_downsample_bwd_bt(N.fp(src), N.fp(w), N.fp(gb), ctx.ds, T, B, C,
                                                  N.fp(d_src), N.fp(dw), N.stream()), "s2t_downsample_bwd_bt")
            return d_src, dw, None, None
        g = g.contiguous().float()
        N.check(N.lib().s2t_downsample_bwd(N.fp(src), N.fp(w), N.fp(g), ctx.ds, T, B, C,
                                           N.fp(d_src), N.fp(dw), N.stream()), "s2t_downsample_bwd")
        return d_src, dw, None, None


def simple_downsample(src, bias, ds, batch_major=False):
    """SimpleDownsample (reference zipformer.py:1653-1695).  HIP: zip_glue.hip.  batch_major: the
    result is STORED (B, T', C) and returned as its (T', B, C) view (values unchanged)."""
    if src.is_cuda and src.dim() == 3 and 1 <= ds <= 8:
        return _Downsample.apply(src, bias.softmax(dim=0), ds, batch_major)
    T, B, C = src.shape
    dT = (T + ds - 1) // ds
    pad = dT * ds - T
    if pad:
        src = torch.cat((src, src[T - 1:].expand(pad, B, C)), dim=0)
    w = bias.softmax(dim=0).reshape(1, ds, 1, 1)
    return (src.reshape(dT, ds, B, C) * w).sum(dim=1)


def simple_upsample(src, up, out_len):
    T, B, C = src.shape
    return src.unsqueeze(1).expand(T, up, B, C).reshape(T * up, B, C)[:out_len]


class _BypassUp(torch.autograd.Function):
    """out[t] = orig[t] + (src[t // up] - orig[t]) * scale: SimpleUpsample + the out_combiner
    bypass of a downsampled stack without materialising the upsampled tensor."""

    @staticmethod
    def forward(ctx, orig, src, scale, up):
        orig, src = orig.contiguous().float(), src.contiguous().float()
        scale = scale.contiguous().float()
        T, B, C = orig.shape
        out = torch.empty_like(orig)
        N.PROF[0] and N.profile_note("s2t_bypass_up_fwd", 4.0 * (2 * orig.numel() + src.numel()))
        N.check(N.lib().s2t_bypass_up_fwd(N.fp(orig), N.fp(src), N.fp(scale), up, T, B, C,
                                          N.fp(out), N.stream()), "s2t_bypass_up_fwd")
        ctx.save_for_backward(orig, src, scale)
        ctx.up = up
        return out

    @staticmethod
    def backward(ctx, g):
        orig, src, scale = ctx.saved_tensors
        T, B, C = orig.shape
        g = g.contiguous().float()
        d_orig, d_src = torch.empty_like(orig), torch.empty_like(src)
        d_scale = torch.zeros_like(scale)
        N.PROF[0] and N.profile_note("s2t_bypass_up_bwd", 4.0 * (3 * orig.numel() + 2 * src.numel()))
        N.check(N.lib().s2t_bypass_up_bwd(N.fp(orig), N.fp(src), N.fp(scale), N.fp(g), ctx.up, T, B,
                                          C, N.fp(d_orig), N.fp(d_src), N.fp(d_scale), N.stream()),
                "s2t_bypass_up_bwd")
        return d_orig, d_src, d_scale, None


def bypass_upsampled(orig, src, scale, up):
    """out_combiner(orig, upsample(src)[:T]) of DownsampledZipformer2Encoder.forward
    (reference zipformer.py:1253-1283).  HIP: zip_glue.hip."""
    T = orig.shape[0]
    if (scale.dim() == 1 and orig.is_cuda and orig.shape[-1] % 4 == 0 and src.shape[1:] == orig.shape[1:]
            and src.shape[0] == (T + up - 1) // up and orig.data_ptr() % 16 == 0
            and src.data_ptr() % 16 == 0):
        return _BypassUp.apply(orig, src, scale, up)
    return bypass_combine(orig, simple_upsample(src, up, T), scale)


# ------------------------------------------------------------------ linear layers
def _wgrad_ok(g2, a2):
    return (g2.dtype == torch.float32 and a2.dtype == torch.float32 and g2.stride(1) == 1
            and a2.stride(1) == 1 and g2.shape[1] % 2 == 0 and a2.shape[1] % 2 == 0
            and g2.stride(0) % 2 == 0 and a2.stride(0) % 2 == 0 and g2.shape[1] >= 2
            and a2.shape[1] >= 2 and g2.data_ptr() % 8 == 0 and a2.data_ptr() % 8 == 0)


def linear_wgrad(g2, a2, want_bias):
    """g2 (R,N), a2 (R,M) -> dW (N,M) = g2^T a2 and db (N) = column sums (or None).  The
    split-row MFMA kernel serves the tall shapes (R >> N,M: a handful of output tiles, where a
    library GEMM leaves most of the chip idle); big outputs stay with hipBLASLt."""
    _dev(g2, a2)
    R, Nf = g2.shape
    Mf = a2.shape[1]
    if _tn_ok(g2) and _tn_ok(a2) and R >= 2048:
        # TN MFMA GEMM (accumulating): one zeroed (N+1, M) buffer receives dW and, in its last
        # row's first N entries, nothing -- the bias gradient has its own zeroed vector
        dW = torch.zeros((Nf, Mf), dtype=torch.float32, device=g2.device)
        db = torch.zeros((Nf,), dtype=torch.float32, device=g2.device) if want_bias else None
        gemm_tn(g2, a2, dW, db)
        return dW, db
    tiles = ((Nf + 63) // 64) * ((Mf + 63) // 64)
    if R >= 3000 and tiles <= WGRAD_MAX_TILES and _wgrad_ok(g2, a2):
        dW = torch.empty((Nf, Mf), dtype=torch.float32, device=g2.device)
        db = torch.empty((Nf,), dtype=torch.float32, device=g2.device) if want_bias else None
        ws = torch.empty(N.lib().s2t_linear_wgrad_workspace_floats(R, Nf, Mf), dtype=torch.float32,
                         device=g2.device)
        N.PROF[0] and N.profile_note("s2t_linear_wgrad", 4.0 * (g2.numel() + a2.numel() + dW.numel()))
        N.check(N.lib().s2t_linear_wgrad(N.raw(g2, torch.float32), g2.stride(0),
                                         N.raw(a2, torch.float32), a2.stride(0), R, Nf, Mf,
                                         N.fp(dW), N.fp(db), 0, N.fp(ws), N.stream()),
                "s2t_linear_wgrad")
        return dW, db
    return _wgrad_splitk(g2, a2), (g2.sum(dim=0) if want_bias else None)


WGRAD_MAX_TILES = 64   # measured crossover against hipBLASLt (tools/bench_kernels.py wgrad)


def _tn_ok(t):
    return (t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0
            and t.shape[1] % 4 == 0 and t.shape[0] >= 4 and t.data_ptr() % 16 == 0)


class _Side:
    """Weight-gradient GEMMs of backward run on the library's side stream (streams.hip): they
    only feed the optimizer, so they overlap the data-gradient chain.  Operands are kept alive
    until the join, which autograd runs as an end-of-backward callback."""
    enabled = os.environ.get("S2T_WGRAD_STREAM", "1") == "1"
    handle = None
    keep = []
    queued = False


def side_stream_handle():
    """Raw handle of the side stream if weight-gradient work may be in flight on it."""
    return _Side.handle


def _side_join():
    if _Side.handle is not None:
        N.check(N.lib().s2t_stream_order(_Side.handle, N.stream()), "s2t_stream_order(join)")
    _Side.keep.clear()
    _Side.queued = False


def side_sync():
    """Orders the current stream after whatever the side stream holds and drops the operand
    references.  The trainer calls it at the start of every step and again before the optimizer:
    autograd skips its end-of-backward callbacks when backward raises (e.g. a batch skipped after
    an out-of-memory error), which would otherwise leave `queued` set -- later backwards would
    then never join the side stream and the optimizer could read gradients still being written."""
    if _Side.handle is not None:
        N.check(N.lib().s2t_stream_order(_Side.handle, N.stream()), "s2t_stream_order(sync)")
    _Side.keep.clear()
    _Side.queued = False


def _side_launch_stream(*tensors):
    """Orders the side stream after the work enqueued so far on the current stream and returns
    its handle; falls back to the current stream outside a backward pass."""
    if not _Side.enabled:
        return None
    if _Side.handle is None:
        h = N.lib().s2t_side_stream()
        if not h:
            _Side.enabled = False
            return None
        _Side.handle = ctypes.c_void_p(h)
    if not _Side.queued:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_side_join)
        except RuntimeError:                       # not inside a backward pass
            return None
        _Side.queued = True
    N.check(N.lib().s2t_stream_order(N.stream(), _Side.handle), "s2t_stream_order(fork)")
    _Side.keep.append(tensors)
    return _Side.handle


_EXT = {}
_WGRAD_SIDE = os.environ.get("S2T_WGRAD_SIDE_MORE", "1") == "1"


# which of the FRONTEND's parameter gradients fork to the side stream (bits: 1 = the 7x7 depthwise,
# 2 = the 3x3 convs' implicit-im2col products, 4 = the 600 k-row Linears and the output Linear).  Its backward is a run of
# HBM-bound passes over 300-900 MB maps: two of them at once share the same bandwidth.
_FRONT_SIDE = int(os.environ.get("S2T_FRONT_W_SIDE", "7"))


def side_param_grads(params, compute, keep=(), allow=True):
    """Parameter gradients that only feed the optimizer, off the data-gradient chain: when every
    parameter of `params` (None entries allowed) is a leaf whose `.grad` is a flat-store view and a
    backward pass is running, `compute()` -- the launches that produce the gradients, in the
    parameters' shapes -- runs with the library's SIDE stream as torch's current stream (ordered
    after the work enqueued so far), its results are added into the `.grad` views there, and a
    list of None is returned for autograd; operands in `keep` stay referenced until the join.
    Otherwise `compute()` runs on the current stream and its tensors are returned."""
    ok = allow and _WGRAD_SIDE and _Side.enabled
    if ok:
        for q in params:
            if q is None:
                continue
            if not (q.is_leaf and flat.owned(q) and q.grad is not None and q.grad.is_contiguous()):
                ok = False
                break
    h = _side_launch_stream(*keep) if ok else None
    if h is None:
        return compute()
    ext = _EXT.get(h.value)
    if ext is None:
        ext = _EXT[h.value] = torch.cuda.ExternalStream(h.value)
    with torch.cuda.stream(ext):
        grads = compute()
        for q, gq in zip(params, grads):
            if q is not None and gq is not None:
                q.grad.add_(gq.reshape(q.grad.shape) if gq.shape != q.grad.shape else gq)
    _Side.keep.append(tuple(grads))
    return [None] * len(grads)


def gemm_tn(g2, a2, out, colsum=None, pro=0, stream=None):
    """out (N,M) += g2^T act(a2); colsum (N) += column sums of g2.   HIP: gemm.hip mode TN."""
    R, Nf = g2.shape
    Mf = a2.shape[1]
    N.PROF[0] and N.profile_note("s2t_gemm_f32", 4.0 * (g2.numel() + a2.numel() + out.numel()),
                   2.0 * R * Nf * Mf)
    N.check(N.lib().s2t_gemm_f32(2, N.raw(g2, torch.float32), g2.stride(0),
                                 N.raw(a2, torch.float32), a2.stride(0), N.fp(out), out.stride(0),
                                 Nf, Mf, R, None, None, 0, None, 0, 0, 0, int(pro),
                                 N.fp(colsum), 0, stream if stream is not None else N.stream()),
            "s2t_gemm_f32(TN)")


def wgrad_into(wparam, bparam, g2, a2, pro=0, notify=False):
    """Accumulates dW = g2^T act(a2) (and db) DIRECTLY into wparam.grad / bparam.grad when those
    are the flat-store views (speech2text_amd.flat): one launch, no temporary, no autograd
    accumulate kernel.  Returns False when that is not possible (the caller then returns the
    gradients as tensors).  notify: the caller is not an autograd node of these parameters (the
    layer executor), so no post-accumulate hook will follow -- tell the gradient reducer here."""
    if not (wparam.is_leaf and flat.owned(wparam) and _tn_ok(g2) and _tn_ok(a2)):
        return False
    wg = wparam.grad
    if wg is None or not wg.is_contiguous() or wg.dim() < 2:
        return False
    if wg.dim() > 2:                                 # (N, M, 1) weight of a 1x1 convolution
        wg = wg.view(wg.shape[0], -1)
    bg = None
    if bparam is not None:
        if not (bparam.is_leaf and flat.owned(bparam)):
            return False
        bg = bparam.grad
        if bg is None or not bg.is_contiguous():
            return False
    # (main stream: forking to the side stream per Linear costs the host more than the overlap
    # gives back -- measured on the conformer step; the grouped per-layer launch below does fork,
    # and so do the few products big enough to matter: the frontend's 600 k-row maps)
    st = None
    if _WGRAD_SIDE and (_FRONT_SIDE & 4) and 2.0 * g2.shape[0] * g2.shape[1] * a2.shape[1] >= 2.0e10:
        st = _side_launch_stream(g2, a2)
    gemm_tn(g2, a2, wg, bg, pro, stream=st)
    if notify:
        flat.grad_written(wparam)
        if bparam is not None:
            flat.grad_written(bparam)
    return True


_BMM_OWN = os.environ.get("S2T_BMM_OWN", "1") == "1"


def batched_matmul(mode, a, b, own_tn=False):
    """Batch of independent fp32 products (the nonlinear attention's attn_weights[0] @ x and its
    two gradients, reference zipformer.py:2438-2483), contiguous 3-D operands:
      mode 0: a (n,M,K) b (n,N,K) -> a @ b^T;  mode 1: a (n,M,K) b (n,K,N) -> a @ b;
      mode 2: a (n,K,M) b (n,K,N) -> a^T @ b.
    Modes 0 / 1: one launch of our bf16x3 kernel (s2t_gemm_f32_batched); torch.bmm (rocBLAS) for
    mode 2 and for the shapes the kernel's alignment rules refuse (T = 495, 62)."""
    n = a.shape[0]
    if mode == 0:
        M, K, Nn = a.shape[1], a.shape[2], b.shape[1]
    elif mode == 1:
        M, K, Nn = a.shape[1], a.shape[2], b.shape[2]
    else:
        M, K, Nn = a.shape[2], a.shape[1], b.shape[2]
    # measured at the C3 shapes (tools/debug/bmm_test.py, 64 x 248 x 248 x 192): a @ b 26 us against
    # 55, a @ b^T 21 against 69; the a^T @ b form (split contraction + atomics) 120 against 55 --
    # that one stays with the library, as do the shapes the kernel's alignment rules refuse
    # own_tn: the a^T @ b form on our kernel as well -- for SHORT contractions (the simple loss's
    # W^T @ exp(lm): K = S + 1 = 51, one slice, no split) it needs no second pass; its tiles are ADDED
    # to the output, which therefore starts as zeros
    tn_ok = own_tn and mode == 2 and M % 4 == 0 and Nn % 4 == 0
    if (_BMM_OWN and (mode != 2 or tn_ok) and (mode == 2 or K % 4 == 0) and (mode == 0 or Nn % 4 == 0)
            and min(M, Nn, K) >= 4 and a.is_cuda and a.dtype is torch.float32 and b.dtype is torch.float32
            and a.is_contiguous() and b.is_contiguous() and n > 0):
        out = (torch.zeros if mode == 2 else torch.empty)((n, M, Nn), dtype=torch.float32, device=a.device)
        N.PROF[0] and N.profile_note("s2t_gemm_f32_batched", 4.0 * (a.numel() + b.numel() + out.numel()),
                                     2.0 * n * M * Nn * K)
        rc = N.lib().s2t_gemm_f32_batched(mode, N.fp(a), a.stride(1), a.stride(0), N.fp(b), b.stride(1),
                                          b.stride(0), N.fp(out), Nn, M * Nn, M, Nn, K, n, N.stream())
        if rc == 0:
            return out
        if rc != -2:
            N.check(rc, "s2t_gemm_f32_batched")
    if a.is_cuda and a.dtype is torch.float32 and b.dtype is torch.float32 and a.is_contiguous() \
            and b.is_contiguous() and n > 0:
        # the library's strided-batch product through our C ABI (the call csrc/zip_layer.hip makes)
        out = torch.empty((n, M, Nn), dtype=torch.float32, device=a.device)
        ws = _lt_workspace(a.device)
        N.PROF[0] and N.profile_note("s2t_bmm_lt", 4.0 * (a.numel() + b.numel() + out.numel()), 2.0 * n * M * Nn * K)
        rc = N.lib().s2t_bmm_lt(mode, N.fp(a), N.fp(b), N.fp(out), n, M, Nn, K, ws.data_ptr(), ws.numel(),
                                N.stream())
        if rc == 0:
            return out
        if rc != -2:
            N.check(rc, "s2t_bmm_lt")
    if mode == 0:
        return torch.bmm(a, b.transpose(1, 2))
    if mode == 1:
        return torch.bmm(a, b)
    return torch.bmm(a.transpose(1, 2), b)


class TnProblem(ctypes.Structure):
    """Mirror of S2tTnProblem (include/s2t_mi355.h)."""
    _fields_ = [("A", ctypes.c_void_p), ("lda", ctypes.c_long), ("B", ctypes.c_void_p),
                ("ldb", ctypes.c_long), ("C", ctypes.c_void_p), ("ldc", ctypes.c_long),
                ("M", ctypes.c_int), ("N", ctypes.c_int), ("K", ctypes.c_int),
                ("colsum", ctypes.c_void_p), ("alpha", ctypes.c_float)]


def wgrad_group(items):
    """items: [(weight, bias | None, g2 (R,N), a2 (R,M)[, alpha])] -- the weight / bias gradients of a whole
    layer, accumulated into the flat gradient views by ONE grouped TN launch on the side stream
    (gemm.hip s2t_gemm_tn_grouped).  The caller is not an autograd node of these parameters (the
    layer executor), so the gradient reducer is told here.  Operands that do not meet the
    kernel's layout rules fall back to wgrad_into / a library GEMM one by one."""
    ok = []
    for it in items:
        w, b, g2, a2 = it[:4]
        alpha = it[4] if len(it) > 4 else 1.0
        wg = w.grad
        if (_tn_ok(g2) and _tn_ok(a2) and flat.owned(w) and wg is not None and wg.is_contiguous()
                and (b is None or (flat.owned(b) and b.grad is not None))):
            ok.append((w, b, g2, a2, alpha))
            continue
        if alpha != 1.0 or not wgrad_into(w, b, g2, a2, 0, notify=True):
            dw, db = linear_wgrad(g2, a2, b is not None)
            w.grad.add_(dw.view(w.shape), alpha=alpha)
            flat.grad_written(w)
            if b is not None:
                b.grad.add_(db, alpha=alpha)
                flat.grad_written(b)
    n = len(ok)
    if n == 0:
        return
    arr = (TnProblem * n)()
    nbytes = flops = 0.0
    for q, (w, b, g2, a2, alpha) in zip(arr, ok):
        q.alpha = alpha
        R, Nf = g2.shape
        Mf = a2.shape[1]
        q.A, q.lda = g2.data_ptr(), g2.stride(0)
        q.B, q.ldb = a2.data_ptr(), a2.stride(0)
        q.C, q.ldc = w.grad.data_ptr(), Mf
        q.M, q.N, q.K = Nf, Mf, R
        q.colsum = None if b is None else b.grad.data_ptr()
        nbytes += 4.0 * (g2.numel() + a2.numel() + Nf * Mf)
        flops += 2.0 * R * Nf * Mf
    N.PROF[0] and N.profile_note("s2t_gemm_tn_grouped", nbytes, flops)
    st = _side_launch_stream(ok)
    N.check(N.lib().s2t_gemm_tn_grouped(n, ctypes.cast(arr, ctypes.c_void_p),
                                        st if st is not None else N.stream()),
            "s2t_gemm_tn_grouped")
    for w, b, _, _, _ in ok:
        flat.grad_written(w)
        if b is not None:
            flat.grad_written(b)


_LT_WS = {}


def _lt_workspace(dev):
    ws = _LT_WS.get(dev)
    if ws is None:
        ws = torch.empty(32 << 20, dtype=torch.uint8, device=dev)
        _LT_WS[dev] = ws
    return ws


def _rows(t):
    """(..., K) -> (R, K) view with unit inner stride (copy only if the layout forces one)."""
    t2 = t.reshape(-1, t.shape[-1])
    if t2.stride(1) != 1 or t2.stride(0) < t2.shape[1] or t2.dtype != torch.float32:
        t2 = t2.contiguous().float()
    return t2


_ACTK = {None: 0, "swoosh_l": 1, "swoosh_r": 2, "add": 3}
X3P = {"on": os.environ.get("S2T_X3P", "1") == "1", "calls": 0, "tile": 0,
       "tune": os.environ.get("S2T_X3P_TUNE", "1") == "1",
       "margin": float(os.environ.get("S2T_X3P_MARGIN", "0.97"))}


def _vp(t):
    return None if t is None else t.data_ptr()       # (int: the entry points carry argtypes)


def x3p_matmul(mode, x2, w2, bias=None, resid2=None, act_src=None, act_kind=None, act2=None, tile=0,
               resid_b=None, pp=None, bal=None, cls=None):
    """The same products as lt_matmul on our bf16x3 kernel with pre-split weight pieces
    (csrc/gemm_x3p.hip, planes.py): mode 0: x2 (R,K) w2 (N,K)^T (+bias) -> (R,N); mode 1: x2 (R,N)
    w2 (N,K) -> (R,K); then (* act'(act_src)) (+ resid2); with act2 a second output act2(result).
    Returns None when the weight has no pieces (not in a FlatStore / shape outside the kernel's
    rules) -- the caller takes the library path."""
    if pp is None:
        pp = planes.pieces(w2, mode)
        if pp is None:
            return None
    R = x2.shape[0]
    Nf, Kf = w2.shape
    cols, inner = (Nf, Kf) if mode == 0 else (Kf, Nf)
    for t in (x2, bias, resid2, act_src, resid_b):       # (alignment is checked by the entry point: -2)
        if t is not None and (t.dtype is not torch.float32 or not t.is_cuda):
            return None
    out = torch.empty((R, cols), dtype=torch.float32, device=x2.device)
    out2 = torch.empty_like(out) if act2 is not None else None
    if R == 0:
        return out if out2 is None else (out, out2)
    if N._Prof.target is not None:
        extra = sum(t is not None for t in (resid2, act_src, out2, resid_b))
        N.profile_note("s2t_gemm_x3p", 4.0 * R * (Nf + Kf + extra * cols)
                       + 2.0 * gemm_arith(cls if cls is not None else mode) * Nf * Kf,
                       2.0 * R * Nf * Kf)             # (_native.SELF_NOTING: not behind N.PROF)
    prev_cls = N.lib().s2t_gemm_class_set(cls if cls is not None else mode)
    try:
        return _x3p_launch(mode, x2, w2, bias, resid2, act_src, act_kind, act2, tile, resid_b, pp, bal,
                           R, Nf, Kf, cols, inner, out, out2)
    finally:
        N.lib().s2t_gemm_class_set(prev_cls)


def _x3p_launch(mode, x2, w2, bias, resid2, act_src, act_kind, act2, tile, resid_b, pp, bal, R, Nf, Kf, cols,
                inner, out, out2):
    if bal is not None:
        # Balancer on act_src in the epilogue: its column statistics first (one read of act_src)
        if act_src is None or bias is not None or out2 is not None or resid_b is not None or cols > 1024:
            return None
        stats = torch.zeros(4096, dtype=torch.float32, device=x2.device)      # sums | squares | a | b
        N.PROF[0] and N.profile_note("s2t_balancer_stats", 4.0 * act_src.numel())
        N.PROF[0] and N.profile_note("s2t_gemm_x3p_bal", 4.0 * R * (Nf + Kf + cols) + 6.0 * Nf * Kf, 2.0 * R * Nf * Kf)
        N.check(N.lib().s2t_balancer_stats(_vp(act_src), act_src.stride(0), R, cols, _vp(stats), N.stream()),
                "s2t_balancer_stats")
        rc = N.lib().s2t_gemm_x3p_bal(_vp(x2), x2.stride(0), ctypes.c_void_p(pp), cols, inner, _vp(out), cols, R,
                                      _vp(resid2), 0 if resid2 is None else resid2.stride(0), _vp(act_src),
                                      act_src.stride(0), _ACTK[act_kind], tile or X3P["tile"], _vp(stats),
                                      bal[0], bal[1], bal[2], bal[3], bal[4], N.stream())
    else:
        rc = N.lib().s2t_gemm_x3p(_vp(x2), x2.stride(0), ctypes.c_void_p(pp), cols, inner, _vp(out), cols, R,
                                  _vp(bias), _vp(resid2), 0 if resid2 is None else resid2.stride(0),
                                  _vp(act_src), 0 if act_src is None else act_src.stride(0),
                                  _ACTK[act_kind], _vp(out2), cols, _ACTK[act2], _vp(resid_b),
                                  0 if resid_b is None else resid_b.stride(0), tile or X3P["tile"],
                                  N.stream())
    if rc == -2:
        return None
    N.check(rc, "s2t_gemm_x3p")
    X3P["calls"] += 1
    return out if out2 is None else (out, out2)


_PLANS = {}          # shape bucket + epilogue -> ("lt", 0) | ("x3p", tile)
_BASE = {}           # shape bucket -> (library ms, best own ms | None, its tile): timed once, on the plain product
# tile / occupancy candidates timed per shape bucket: 100 w + tile = the register-staged form at w
# workgroups per CU; 2000 + tile = the LDS-DMA form (weight pieces global -> LDS directly) at 3 / 4 /
# 4 workgroups per CU (S2T_X3P_DMA=1 adds them: same-box A/B at C3 38.6 ms/step with and without)
_X3P_TILES = (222, 321, 312, 411) + ((2022, 2021, 2012) if os.environ.get("S2T_X3P_DMA", "0") == "1" else ())
# two-piece arithmetic (S2T_GEMM_ARITH=2): the same tiles of the register-staged form, the LDS-DMA form
# at 16-deep stages (2000 +) and at 32-deep barrier intervals (2200 +)
_X3P_TILES2 = tuple(int(t) for t in os.environ.get(
    "S2T_X3P_TILES2", "222,321,312,411,2022,2021,2012,2222,2221,2212,2211").split(","))
PLAN_STATS = {"timed": 0}


CLS_F, CLS_D, CLS_W, CLS_S = 0, 1, 2, 3     # classes of product (include/s2t_mi355.h s2t_gemm_arith_of)


def gemm_arith(cls=-1):
    """Pieces per fp32 operand of the bf16 matrix-core GEMMs of class `cls` (forward, data gradient,
    weight gradient, statistics; -1: the base value), as the library reads it for the next call
    (3: bf16x3, six products, fp32-exact; 2: bf16x2, three products) -- include/s2t_mi355.h."""
    return int(N.lib().s2t_gemm_arith_of(cls))


def gemm_arith_name(a=None):
    return {3: "bf16x3/6", 2: "bf16x2/3"}[gemm_arith() if a is None else a]


def gemm_arith_policy():
    """(name, per-class dict): name = 'bf16x2/3' | 'bf16x3/6' when the forward, data-gradient and
    weight-gradient products run the same arithmetic (the statistics class is listed in the dict), else
    e.g. 'F bf16x3/6, D bf16x2/3, W bf16x2/3'."""
    v = {n: gemm_arith(c) for n, c in zip("FDWS", (CLS_F, CLS_D, CLS_W, CLS_S))}
    classes = {n: gemm_arith_name(a) for n, a in v.items()}
    if v["F"] == v["D"] == v["W"]:
        return gemm_arith_name(v["F"]), classes
    return ", ".join(f"{n} {classes[n]}" for n in "FDW"), classes


class gemm_class:
    """with gemm_class(CLS_D): ... -- the class of the x3p / NT / NN / batched products issued inside
    (the calling thread's, s2t_gemm_class_set)."""

    def __init__(self, cls):
        self.cls = cls

    def __enter__(self):
        self.prev = N.lib().s2t_gemm_class_set(self.cls)

    def __exit__(self, *exc):
        N.lib().s2t_gemm_class_set(self.prev)
        return False


def _half_octave(m):
    """floor(2 log2 m), in integers."""
    b = max(1, m).bit_length() - 1
    return 2 * b + (1 if m * m >= (1 << (2 * b + 1)) else 0)


def _time_call(fn, reps=4, groups=2):
    """ms per call: the better of `groups` back-to-back groups of `reps` launches after one untimed
    one (a single group picked a different tile from run to run often enough to move the step by
    0.2-0.3 ms: eleven candidates a few per cent apart, timed once each)."""
    fn()
    best = None
    for _ in range(groups):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        best = ms if best is None or ms < best else best
    return best


_BAL_EPI = os.environ.get("S2T_BAL_EPI", "1") == "1"


def lt_matmul(mode, x2, w2, bias=None, resid2=None, act_src=None, act_kind=None, act2=None,
              resid_b=None, bal=None):
    """Forward / data-gradient product of a Linear with its elementwise neighbours:
      mode 0: x2 (R,K) w2 (N,K)^T (+bias) -> (R,N);  mode 1: x2 (R,N) w2 (N,K) -> (R,K);
      then (* act'(act_src), act_kind "swoosh_l" | "swoosh_r") (+ resid2) (+ resid_b); with act2
      "swoosh_l" | "swoosh_r" a second output act2(result) is returned as well; act2 "add": the
      second output is result + resid_b (the result itself then excludes resid_b).
    Served by our bf16x3 kernel with pre-split weight pieces (s2t_gemm_x3p, epilogue-fused) or by
    the plan cache of s2t_linear_lt (+ a separate activation pass) -- whichever was faster when the
    shape bucket {mode, half-octave of R, N, K} was first seen (the plain product timed once, on the
    call's own operands, on an otherwise idle chip; epilogue variants are decided from those times).  Weights outside a FlatStore always take the latter."""
    fused = act_src is not None or act2 is not None or resid_b is not None
    # bal = Balancer.cfg(2) of a Balancer that sits on act_src (the activation's input) and fires this
    # call: result = balancer(act'(act_src) * product) -- folded into our kernel's epilogue
    # (s2t_gemm_x3p_bal), or the two-pass s2t_balancer_bwd after the library product
    assert bal is None or (act_src is not None and bias is None and resid2 is None and act2 is None
                           and resid_b is None)
    if bal is not None and not _BAL_EPI:           # (A/B switch: the separate two-pass update)
        return balancer_backward(act_src, lt_matmul(mode, x2, w2), *bal, swoosh_l=(act_kind == "swoosh_l"))

    def lib():
        if bal is not None:
            y = _lt_matmul_lib(mode, x2, w2)
            return balancer_backward(act_src, y, *bal, swoosh_l=(act_kind == "swoosh_l"))
        y = _lt_matmul_lib(mode, x2, w2, bias, None if act_src is not None else resid2)
        if act_src is not None:
            y = swoosh_backward(act_src, y, act_kind == "swoosh_l")
            if resid2 is not None:
                y = y + resid2
        if act2 == "add":
            return y, y + resid_b
        if resid_b is not None:
            y = y + resid_b
        if act2 is not None:
            return y, swoosh_forward(y, act2 == "swoosh_l")
        return y

    pp = planes.pieces(w2, mode) if (X3P["on"] and x2.shape[0]) else None
    if pp is None:
        return lib()
    arith = N.lib().s2t_gemm_arith_of(mode)   # (read per call, per class of product: the buckets of the two arithmetics are apart)
    base = (mode | (arith << 4), _half_octave(x2.shape[0]), w2.shape[0], w2.shape[1])
    key = base + (bias is not None, resid2 is not None, act_src is not None, act2, resid_b is not None,
                  bal is not None)
    plan = _PLANS.get(key)
    if plan is None:
        # TIMING happens once per SHAPE bucket, on the plain product (the first call's bias /
        # residual), i.e. during the first step: every shape occurs in every step, but which
        # epilogue a module asks for depends on the step's random Balancer / Whiten decisions, and a
        # variant first seen in step 7 must not stall the pipeline for 25 timed launches.  A new
        # epilogue variant of a timed shape is decided from those two numbers and the bytes the
        # separate passes of the library path would move.
        b = _BASE.get(base)
        if b is None:
            if not X3P["tune"]:
                b = (float("inf"), 0.0, 0)
            else:
                torch.cuda.synchronize()       # side streams idle: candidates are compared alone
                t_lib = _time_call(lambda: _lt_matmul_lib(mode, x2, w2, bias, resid2))
                t_own, tile = None, 0
                for t in (_X3P_TILES2 if arith == 2 else _X3P_TILES):
                    if x3p_matmul(mode, x2, w2, bias, resid2, tile=t) is None:
                        break
                    ms = _time_call(lambda: x3p_matmul(mode, x2, w2, bias, resid2, tile=t))
                    if t_own is None or ms < t_own:
                        t_own, tile = ms, t
                b = (t_lib, t_own, tile)
                PLAN_STATS["timed"] += 1
            _BASE[base] = b
            # the native layer executor (csrc/zip_layer.hip) decides from the same numbers
            N.lib().s2t_zl_plan_put(base[0], base[1], base[2], base[3], float(b[0]),
                                    -1.0 if b[1] is None else float(b[1]), int(b[2]))
        t_lib, t_own, tile = b
        if t_own is None:
            plan = ("lt", 0)
        else:
            rc = float(x2.shape[0]) * (w2.shape[0] if mode == 0 else w2.shape[1])
            pass_ms = 4.0e-3 + 12.0 * rc / 3.0e9                 # one elementwise pass: 2 reads + 1 write at 3 TB/s
            n_pass = ((act_src is not None) + (act_src is not None and resid2 is not None)
                      + (resid_b is not None) + (act2 in ("swoosh_l", "swoosh_r")) + (bal is not None))
            n_ops = (act_src is not None) + (resid_b is not None) + (act2 is not None) + (bal is not None)
            cost_lt = t_lib * (1.0 if fused else X3P["margin"]) + n_pass * pass_ms
            cost_own = t_own + n_ops * 4.0 * rc / 3.0e9          # each extra operand / output: one more stream
            plan = ("x3p", tile) if cost_own < cost_lt else ("lt", 0)
        if os.environ.get("S2T_PLAN_DUMP"):
            print(f"[s2t plan] mode {mode} R {x2.shape[0]} N {w2.shape[0]} K {w2.shape[1]} "
                  f"bias {bias is not None} resid {resid2 is not None} fused {fused}: {plan} "
                  f"(lt {1e3 * t_lib:.1f} us, own {'-' if t_own is None else round(1e3 * t_own, 1)} us)", flush=True)
        _PLANS[key] = plan
    if plan[0] == "x3p":
        y = x3p_matmul(mode, x2, w2, bias, resid2, act_src, act_kind, act2, plan[1], resid_b, pp,
                       bal=bal)
        if y is not None:
            return y
    return lib()


def _lt_matmul_lib(mode, x2, w2, bias=None, resid2=None, out_shape=None):
    """mode 0: x2 (R,K) w2 (N,K)^T (+bias) (+resid2) -> (R,N);  mode 1: x2 (R,N) w2 (N,K) (+resid2)
    -> (R,K).  One hipBLASLt launch with the bias / residual in the epilogue (gemm_lib.hip);
    torch.matmul when the library has no algorithm for the shape."""
    R = x2.shape[0]
    Nf, Kf = w2.shape
    cols = Nf if mode == 0 else Kf
    out = torch.empty((R, cols), dtype=torch.float32, device=x2.device)
    if R == 0:
        return out
    w2 = w2 if (w2.stride(1) == 1 and w2.stride(0) >= w2.shape[1]) else w2.contiguous()
    ws = _lt_workspace(x2.device)
    prev_cls = N.lib().s2t_gemm_class_set(mode)    # (the plan cache may pick our NT / NN kernel: forward / data-gradient class)
    try:
        return _lt_lib_launch(mode, x2, w2, bias, resid2, out, ws, R, Nf, Kf, cols)
    finally:
        N.lib().s2t_gemm_class_set(prev_cls)


def _lt_lib_launch(mode, x2, w2, bias, resid2, out, ws, R, Nf, Kf, cols):
    N.PROF[0] and N.profile_note("s2t_linear_lt", 4.0 * (R * (Nf + Kf) + Nf * Kf + (R * cols if resid2 is not None else 0)),
                   2.0 * R * Nf * Kf)
    rc = N.lib().s2t_linear_lt(mode, N.raw(x2, torch.float32), x2.stride(0),
                               N.raw(w2, torch.float32), w2.stride(0), N.fp(bias),
                               None if resid2 is None else N.raw(resid2, torch.float32),
                               0 if resid2 is None else resid2.stride(0),
                               1.0 if resid2 is not None else 0.0, N.fp(out), cols, R, Nf, Kf,
                               ctypes.c_void_p(ws.data_ptr()), ws.numel(), N.stream())
    if rc == -2:                                   # shape without a library algorithm
        LT_STATS["aten_fallbacks"] += 1
        if LT_STATS["aten_fallbacks"] == 1 or os.environ.get("S2T_LT_LOG_FALLBACK"):
            import warnings
            warnings.warn(f"s2t_linear_lt: no hipBLASLt algorithm for mode {mode} M {R} N {Nf} K {Kf} "
                          f"-- ATen GEMM used (counted in zip_kernels.LT_STATS)")
        y = F.linear(x2, w2, bias) if mode == 0 else x2.matmul(w2)
        return y if resid2 is None else y + resid2
    N.check(rc, "s2t_linear_lt")
    LT_STATS["calls"] += 1
    return out


LT_STATS = {"calls": 0, "aten_fallbacks": 0}


def lt_own_calls():
    """Launches of s2t_linear_lt served by the round-3 bf16x3 kernel (plan cache's choice)."""
    return int(N.lib().s2t_linear_lt_own_calls())


class _Linear(torch.autograd.Function):
    """y = x W^T + b (+ residual): one library GEMM with bias and residual in the epilogue; the
    weight and bias gradients come from one pass of the TN MFMA GEMM over (g, x), accumulated
    in place into the flat gradient buffer when the parameters live in one."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual):
        w2 = weight if weight.dim() == 2 else weight.reshape(weight.shape[0], -1)   # 1x1 conv
        x2 = _rows(x)
        r2 = None if residual is None else _rows(residual)
        y = lt_matmul(0, x2, w2, bias, r2)
        ctx.save_for_backward(x2, weight)
        ctx.params = (weight, bias)
        ctx.xshape = x.shape
        ctx.has_res = residual is not None
        return y.view(x.shape[:-1] + (w2.shape[0],))

    @staticmethod
    def backward(ctx, g):
        x2, weight = ctx.saved_tensors
        wp, bp = ctx.params
        w2 = weight if weight.dim() == 2 else weight.reshape(weight.shape[0], -1)
        g2 = _rows(g)
        dx = lt_matmul(1, g2, w2).view(ctx.xshape) if ctx.needs_input_grad[0] else None
        dres = g if ctx.has_res else None
        if wgrad_into(wp, bp, g2, x2):
            return dx, None, None, dres
        dw, db = linear_wgrad(g2, x2, bp is not None)
        return dx, dw.view(weight.shape), db, dres


class _LinearPass(torch.autograd.Function):
    """(linear(x), x) -- the second output is x itself, to be used as the residual branch of the
    module that starts with this projection.  In backward the residual branch's gradient then
    arrives here and rides in the data-gradient GEMM's epilogue (dx = g W + g_residual) instead
    of costing autograd an extra add kernel per module."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        w2 = weight if weight.dim() == 2 else weight.reshape(weight.shape[0], -1)
        x2 = _rows(x)
        y = lt_matmul(0, x2, w2, bias)
        ctx.save_for_backward(x2, weight)
        ctx.params = (weight, bias)
        ctx.xshape = x.shape
        return y.view(x.shape[:-1] + (w2.shape[0],)), x.view_as(x)

    @staticmethod
    def backward(ctx, g, g_pass):
        x2, weight = ctx.saved_tensors
        wp, bp = ctx.params
        w2 = weight if weight.dim() == 2 else weight.reshape(weight.shape[0], -1)
        g2 = _rows(g)
        dx = None
        if ctx.needs_input_grad[0]:
            r2 = None if g_pass is None else _rows(g_pass)
            dx = lt_matmul(1, g2, w2, None, r2).view(ctx.xshape)
        if wgrad_into(wp, bp, g2, x2):
            return dx, None, None
        dw, db = linear_wgrad(g2, x2, bp is not None)
        return dx, dw.view(weight.shape), db


class _FfnBlock(torch.autograd.Function):
    """y = W2 swooshL(W1 x + b1) + b2 (+ residual), with an optional Balancer on the hidden
    pre-activation -- ConvNeXt's pointwise pair (reference model/layer/subsampling.py:47-57) as ONE
    autograd node: the kept activation leaves the first GEMM's epilogue as a second output, and in
    backward the Swoosh derivative rides in the second GEMM's data-gradient epilogue (or in the
    Balancer's update pass when it fires) -- no elementwise pass over the (rows, hidden) map in
    either direction.  Weight / bias gradients as in _Linear."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, residual, bal_cfg):
        w1m = w1 if w1.dim() == 2 else w1.reshape(w1.shape[0], -1)
        w2m = w2 if w2.dim() == 2 else w2.reshape(w2.shape[0], -1)
        x2 = _rows(x)
        r2 = None if residual is None else _rows(residual)
        h, a = lt_matmul(0, x2, w1m, b1, act2="swoosh_l")
        y = lt_matmul(0, a, w2m, b2, r2)
        ctx.save_for_backward(x2, h, a, w1, w2)
        ctx.params = (w1, b1, w2, b2)
        ctx.bal_cfg, ctx.has_res, ctx.xshape = bal_cfg, residual is not None, x.shape
        return y.view(x.shape[:-1] + (w2m.shape[0],))

    @staticmethod
    def backward(ctx, g):
        x2, h, a, w1, w2 = ctx.saved_tensors
        w1p, b1p, w2p, b2p = ctx.params
        w1m = w1 if w1.dim() == 2 else w1.reshape(w1.shape[0], -1)
        w2m = w2 if w2.dim() == 2 else w2.reshape(w2.shape[0], -1)
        g2 = _rows(g)
        grads2 = (None, None)
        if not wgrad_into(w2p, b2p, g2, a):
            dw, db = linear_wgrad(g2, a, b2p is not None)
            grads2 = (dw.view(w2.shape), db)
        if ctx.bal_cfg is not None:              # Swoosh' AND the Balancer's update in that epilogue
            dh = lt_matmul(1, g2, w2m, act_src=h, act_kind="swoosh_l", bal=ctx.bal_cfg)
        else:                                    # ... or in the data-gradient GEMM's epilogue
            dh = lt_matmul(1, g2, w2m, act_src=h, act_kind="swoosh_l")
        grads1 = (None, None)
        if not wgrad_into(w1p, b1p, dh, x2):
            dw, db = linear_wgrad(dh, x2, b1p is not None)
            grads1 = (dw.view(w1.shape), db)
        dx = lt_matmul(1, dh, w1m).view(ctx.xshape) if ctx.needs_input_grad[0] else None
        return dx, grads1[0], grads1[1], grads2[0], grads2[1], (g if ctx.has_res else None), None


def ffn_block(x, w1, b1, w2, b2, residual=None, balancer_cfg=None):
    """linear(swooshL(balancer(linear(x, w1, b1))), w2, b2) [+ residual] as one autograd node;
    balancer_cfg = Balancer.cfg(2) when the hidden Balancer fires this call, else None."""
    if not x.is_cuda:
        raise RuntimeError("speech2text_amd.ffn_block needs device tensors (HIP path only)")
    return _FfnBlock.apply(x, w1, b1, w2, b2, residual, balancer_cfg)


class _LinearColPerm(torch.autograd.Function):
    """y = x Wp^T + b with Wp = the parameter's columns re-ordered from (c, f) to (f, c) -- the frontend's
    output Linear (reference model/layer/subsampling.py:312-319 flattens (N, C, T, F) c-major; our map is
    channel-LAST, so its rows are f-major).  As its own node so that the weight gradient -- a 30 GFLOP
    product over 31 680 rows that only feeds the optimizer -- leaves the data-gradient chain: side stream,
    added into the flat-store `.grad` views there (side_param_grads).  Through `linear` on a permuted copy
    of the weight it was a main-stream launch of 375 us plus autograd's permute-back and two accumulates."""

    @staticmethod
    def forward(ctx, x, weight, bias, f, c):
        wp = weight.detach().view(-1, c, f).permute(0, 2, 1).reshape(-1, f * c)
        x2 = _rows(x)
        y = lt_matmul(0, x2, wp, bias)
        ctx.save_for_backward(x2, wp)
        ctx.params = (weight, bias)
        ctx.cfg = (x.shape, f, c)
        return y.view(x.shape[:-1] + (wp.shape[0],))

    @staticmethod
    def backward(ctx, g):
        x2, wp = ctx.saved_tensors
        weight, bias = ctx.params
        xshape, f, c = ctx.cfg
        g2 = _rows(g)
        dx = lt_matmul(1, g2, wp).view(xshape) if ctx.needs_input_grad[0] else None

        def compute():
            dwp, db = linear_wgrad(g2, x2, bias is not None)
            return [dwp.view(-1, f, c).permute(0, 2, 1).reshape(weight.shape), db]
        dw, db = side_param_grads((weight, bias), compute, keep=(g2, x2), allow=bool(_FRONT_SIDE & 4))
        return dx, dw, db, None, None


def linear_col_perm(x, weight, bias, f, c):
    """F.linear(x, weight.view(-1, c, f).permute(0, 2, 1).reshape(-1, f * c), bias) for x whose last dim is
    (f, c)-ordered; see _LinearColPerm."""
    if not x.is_cuda:
        raise RuntimeError("speech2text_amd.linear_col_perm needs device tensors (HIP path only)")
    return _LinearColPerm.apply(x, weight, bias, f, c)


def linear_pass(x, weight, bias=None):
    """-> (F.linear(x, weight, bias), alias of x for the residual branch)."""
    return _LinearPass.apply(x, weight, bias)


def linear(x, weight, bias=None, residual=None):
    """F.linear(x, weight, bias) [+ residual] on the GPU."""
    if not x.is_cuda:
        raise RuntimeError("speech2text_amd.linear needs device tensors (HIP path only)")
    return _Linear.apply(x, weight, bias, residual)


# ------------------------------------------------------------------ channel-last frontend convs
def _wgrad_splitk(a, g, chunk=16384):
    """a (R,M), g (R,N) -> a^T g (M,N) for R >> M,N: hipBLASLt gets only a handful of output
    tiles for such shapes, so the reduction dimension is split into a batch of GEMMs."""
    R = a.shape[0]
    S = R // chunk
    if S < 4:
        return a.t().mm(g)
    main = torch.bmm(a[:S * chunk].view(S, chunk, -1).transpose(1, 2),
                     g[:S * chunk].view(S, chunk, -1)).sum(dim=0)
    if S * chunk < R:
        main = main + a[S * chunk:].t().mm(g[S * chunk:])
    return main


def linear_big_m(x, weight, bias):
    return _Linear.apply(x, weight, bias, None)


def _conv3x3_wgrad_implicit(x, g, sh, sw, has_bias):
    """dW (Cout,Cin,3,3) view and db of a 3x3 conv on channel-last x from g (B,Ho,Wo,Cout): the TN
    MFMA GEMM reads the patches straight from x (no im2col matrix)."""
    B, H, W, C = x.shape
    Cout = g.shape[-1]
    dw2 = torch.zeros((Cout, 9 * C), dtype=torch.float32, device=x.device)
    db = torch.zeros((Cout,), dtype=torch.float32, device=x.device) if has_bias else None
    R = g.numel() // Cout
    N.PROF[0] and N.profile_note("s2t_conv3x3_gemm", 4.0 * (x.numel() + g.numel()), 2.0 * R * Cout * 9 * C)
    N.check(N.lib().s2t_conv3x3_gemm(2, N.fp(x), B, H, W, C, sh, sw, Cout, N.fp(g), None, N.fp(dw2),
                                     N.fp(db), N.stream()), "s2t_conv3x3_gemm(wgrad)")
    return dw2.view(Cout, 3, 3, C).permute(0, 3, 1, 2), db


def _implicit_ok(x, Cout, sh=1, sw=1):
    B, H, W, C = x.shape
    rows = B * ((H - 3) // sh + 1) * ((W - 3) // sw + 1)          # patch rows; the kernel wants >= 4
    return x.is_cuda and C % 4 == 0 and Cout % 4 == 0 and x.numel() < (1 << 31) and rows >= 4


_CONV_MAP_FWD = os.environ.get("S2T_CONV_MAP_FWD", "1") == "1"


class _Conv3x3Nhwc(torch.autograd.Function):
    """3x3 conv on channel-last (N,H,W,Cin) as an implicit-im2col GEMM: each patch row is 3
    contiguous runs of 3*Cin floats of x, which the MFMA kernel's operand loader addresses in
    place (s2t_conv3x3_gemm) -- no patch matrix in HBM for the forward or the weight gradient.
    The data gradient is one GEMM into patch space + one col2im gather kernel (zip_front.hip).
    weight keeps nn.Conv2d's (Cout,Cin,3,3) layout.  Channel counts that are not multiples of 4
    use a materialised patch matrix."""

    @staticmethod
    def forward(ctx, x, weight, bias, sh, sw):
        _dev(x, weight)
        x = x.contiguous().float()
        B, H, W, C = x.shape
        Cout = weight.shape[0]
        Ho, Wo = (H - 3) // sh + 1, (W - 3) // sw + 1
        w2 = weight.permute(0, 2, 3, 1).reshape(Cout, 9 * C)                    # cout x (kh,kw,cin)
        ctx.implicit = _implicit_ok(x, Cout, sh, sw)
        pp = None
        if ctx.implicit and _CONV_MAP_FWD and C % 16 == 0 and Cout % 16 == 0:
            # forward on the pre-split bf16x3 GEMM with implicit operands (s2t_gemm_x3p_map: 150 TFLOP/s
            # on the conformer's 256 -> 256 product where the NT kernel below reaches ~ 90)
            pp = planes.adhoc_pieces(w2.detach().contiguous(), 0)
        if pp is not None:
            y = torch.empty((B * Ho * Wo, Cout), dtype=torch.float32, device=x.device)
            amap = RowMap(Ho * Wo, Wo, H * W * C, sh * W * C, sw * C, 0)
            N.check(_x3p_map(x, amap, 3 * C, [0, W * C, 2 * W * C], pp, Cout, y, Cout, None, 0, B * Ho * Wo,
                             None if bias is None else bias.detach()), "s2t_gemm_x3p_map(fwd)")
            ctx.save_for_backward(x, weight)
        elif ctx.implicit:
            y = torch.empty((B * Ho * Wo, Cout), dtype=torch.float32, device=x.device)
            N.PROF[0] and N.profile_note("s2t_conv3x3_gemm", 4.0 * (x.numel() + y.numel()),
                           2.0 * y.numel() * 9 * C)
            N.check(N.lib().s2t_conv3x3_gemm(0, N.fp(x), B, H, W, C, sh, sw, Cout, N.fp(w2),
                                             N.fp(bias), N.fp(y), None, N.stream()),
                    "s2t_conv3x3_gemm")
            ctx.save_for_backward(x, weight)
        else:
            s = x.stride()
            cols = x.as_strided((B, Ho, Wo, 3, 3 * C), (s[0], s[1] * sh, s[2] * sw, s[1], 1)) \
                .reshape(B * Ho * Wo, 9 * C)
            y = lt_matmul(0, cols, w2, bias)
            ctx.save_for_backward(cols, weight)
        ctx.cfg = (B, H, W, C, Ho, Wo, sh, sw, bias is not None)
        ctx.params = (weight, bias)
        return y.view(B, Ho, Wo, Cout)

    @staticmethod
    def backward(ctx, dy):
        xc, weight = ctx.saved_tensors
        B, H, W, C, Ho, Wo, sh, sw, has_bias = ctx.cfg
        Cout = weight.shape[0]
        g = dy.reshape(B * Ho * Wo, Cout)
        g = g if g.is_contiguous() else g.contiguous()
        if ctx.implicit:
            dweight, db = side_param_grads(
                ctx.params, lambda: list(_conv3x3_wgrad_implicit(xc, g, sh, sw, has_bias)), keep=(xc, g),
                allow=bool(_FRONT_SIDE & 2))
        else:
            dwmat, db = linear_wgrad(g, xc, has_bias)                            # (Cout, 9C)
            dweight = dwmat.view(Cout, 3, 3, C).permute(0, 3, 1, 2)
        dx = None
        if ctx.needs_input_grad[0] and ctx.implicit and _CONV_MAP_DGRAD and (sh, sw) == (1, 2) and Cout % 16 == 0 \
                and C >= 16 and g.numel() * 5 < (1 << 31):
            dx = _conv3x3_s12_dgrad_map(g.view(B, Ho, Wo, Cout), weight.detach(), H, W)
        elif ctx.needs_input_grad[0] and ctx.implicit and 9 * C * Cout >= (1 << 18):
            # wide channel counts (the conformer's 256 -> 256 stride-2 conv): the patch-space form
            # below would write and re-read a (rows, 9C) matrix (1.4 GB at C2) -- the library's
            # NHWC implicit-GEMM backward-data kernel serves this one product (measured: C2 step
            # 25.8 ms with the patch-space form)
            gy = g.view(B, Ho, Wo, Cout).permute(0, 3, 1, 2)
            xin = xc.permute(0, 3, 1, 2)
            wcl = weight.contiguous(memory_format=torch.channels_last)
            dxn = torch.ops.aten.convolution_backward(gy, xin, wcl, None, [sh, sw], [0, 0], [1, 1],
                                                      False, [0, 0], 1, [True, False, False])[0]
            dx = dxn.permute(0, 2, 3, 1)
        elif ctx.needs_input_grad[0]:
            w2 = weight.permute(0, 2, 3, 1).reshape(Cout, 9 * C)
            dc = lt_matmul(1, g, w2)                             # (B*Ho*Wo, 3*3*C)
            dx = torch.empty((B, H, W, C), dtype=torch.float32, device=dy.device)
            N.PROF[0] and N.profile_note("s2t_col2im3x3_nhwc", 4.0 * (dc.numel() + dx.numel()))
            N.check(N.lib().s2t_col2im3x3_nhwc(N.fp(dc), B, H, W, C, Ho, Wo, sh, sw, N.fp(dx),
                                               N.stream()), "s2t_col2im3x3_nhwc")
        return dx, dweight, db, None, None


_CONV_MAP_DGRAD = os.environ.get("S2T_CONV_MAP_DGRAD", "1") == "1"
_CONV_MAP_DGRAD_TILE = int(os.environ.get("S2T_CONV_MAP_DGRAD_TILE", "221"))    # (128 x 64 block, 32-deep intervals)


_PAD_BUF = {}


def _conv3x3_s12_dgrad_map(g, wd, H, W):
    """Data gradient of a 3x3 / stride-(1, 2) convolution on a channel-last map (the frontend's 32 -> 128
    convolution, reference model/encoder/zipformer.py Conv2dSubsampling) in GATHER form on the implicit-
    operand GEMM (s2t_gemm_x3p_map): no (rows, 9 C) patch-space matrix, no col2im pass.  Input pixels by
    the parity of their column: an even column w = 2 jc receives the taps kw = 2 (from output column
    jc - 1) and kw = 0 (from jc) of every kh -- two ADJACENT pixels of the zero-bordered gradient, one run
    of 2 Cout floats per kh; an odd column receives kw = 1 (from jc), one run of Cout floats per kh.  One
    launch per class: rows = (b, h, jc), K = 3 runs, output scattered to every second pixel of dx."""
    B, Ho, Wo, Cout = g.shape
    C = wd.shape[1]
    # rows -2 .. Ho + 1, columns -1 .. Wo of the gradient with a zero border: a buffer per shape whose border
    # is zeroed ONCE (only the interior is ever written; the launches below are enqueued on this stream before
    # the next backward's copy) -- F.pad filled 343 MB per step for the sake of 3 MB of border
    key = (B, Ho, Wo, Cout, g.device.index)
    gp = _PAD_BUF.get(key)
    if gp is None:
        _PAD_BUF.clear()
        gp = _PAD_BUF[key] = torch.zeros((B, Ho + 4, Wo + 2, Cout), dtype=torch.float32, device=g.device)
    gp[:, 2:Ho + 2, 1:Wo + 1].copy_(g)
    Wp = Wo + 2
    dx = torch.empty((B, H, W, C), dtype=torch.float32, device=g.device)
    for pw in (0, 1):
        kws = (2, 0) if pw == 0 else (1,)
        # (Cin, 3 * len(kws) * Cout): column (kh, kw, cout) = W[cout, cin, kh, kw]
        bc = torch.cat([wd[:, :, kh, kw].t() for kh in (0, 1, 2) for kw in kws], dim=1).contiguous()
        pp = planes.adhoc_pieces(bc, 0)
        if pp is None:
            raise RuntimeError("conv3x3 (map, stride 1x2): class weight outside the piece kernel's rules")
        Wc = (W - pw + 1) // 2
        amap = RowMap(H * Wc, Wc, (Ho + 4) * Wp * Cout, Wp * Cout, Cout, (2 * Wp + 1) * Cout)
        segoff = [(-kh * Wp - (1 if pw == 0 else 0)) * Cout for kh in (0, 1, 2)]
        cmap = RowMap(H * Wc, Wc, H * W * C, W * C, 2 * C, pw * C)
        N.check(_x3p_map(gp, amap, len(kws) * Cout, segoff, pp, C, dx, C, cmap, dx.numel(), B * H * Wc, None,
                         tile=_CONV_MAP_DGRAD_TILE), "s2t_gemm_x3p_map(dgrad 1x2)")
    return dx


class RowMap(ctypes.Structure):
    """Mirror of S2tRowMap (include/s2t_mi355.h): row r = (b, i, j) -> base + b sb + i sh + j sw floats."""
    _fields_ = [("hw", ctypes.c_int), ("w", ctypes.c_int), ("sb", ctypes.c_long), ("sh", ctypes.c_long),
                ("sw", ctypes.c_long), ("base", ctypes.c_long)]


_CONV_MAP_TILE = int(os.environ.get("S2T_CONV_MAP_TILE", "22"))


def _x3p_map(a, amap, seg, segoff, pp, ncols, out, ldc, cmap, c_elems, M, bias, tile=None):
    tile = _CONV_MAP_TILE if tile is None else tile
    if tile >= 200 and (seg % 32 or N.lib().s2t_gemm_arith() != 2):
        tile -= 200                        # (32-deep intervals: two pieces, segments of whole intervals)
    so = (ctypes.c_long * len(segoff))(*segoff)
    N.PROF[0] and N.profile_note("s2t_gemm_x3p_map", 4.0 * (a.numel() + out.numel()) + 6.0 * ncols * seg * len(segoff),
                                 2.0 * M * ncols * seg * len(segoff))
    return N.lib().s2t_gemm_x3p_map(_vp(a), ctypes.byref(amap), seg, len(segoff), so, ctypes.c_void_p(pp), ncols,
                                    _vp(out), ldc, None if cmap is None else ctypes.byref(cmap), c_elems, M,
                                    _vp(bias), tile, N.stream())


class _Conv3x3S2Map(torch.autograd.Function):
    """3x3 / stride-2 convolution on a channel-last map (N,H,W,C) with wide channels -- the second
    convolution of the conformer's Subsampling (reference model/encoder/conformer.py:47-57, 114-126:
    256 -> 256, 178 GFLOP per pass at C2) -- on the pre-split bf16x3 GEMM with IMPLICIT operands
    (s2t_gemm_x3p_map): no patch matrix in either direction.
      forward       a patch row = 3 runs of 3 C contiguous floats of x;
      weight grad   the split-contraction TN kernel reading the same patches (s2t_conv3x3_gemm);
      data grad     input pixels by parity class (h % 2, w % 2): a pixel of class (0,0) receives 4 taps,
                    (0,1) / (1,0) two, (1,1) one; one launch per class whose rows gather those taps from the
                    zero-bordered output gradient (K = taps * Cout) and write every second pixel of dx."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _dev(x, weight)
        x = x.contiguous().float()
        B, H, W, C = x.shape
        Cout = weight.shape[0]
        Ho, Wo = (H - 3) // 2 + 1, (W - 3) // 2 + 1
        w2 = weight.detach().permute(0, 2, 3, 1).reshape(Cout, 9 * C).contiguous()      # cout x (kh, kw, cin)
        pp = planes.adhoc_pieces(w2, 0)
        if pp is None:
            raise RuntimeError("conv3x3 (map): weight shape outside the piece kernel's rules")
        M = B * Ho * Wo
        y = torch.empty((M, Cout), dtype=torch.float32, device=x.device)
        amap = RowMap(Ho * Wo, Wo, H * W * C, 2 * W * C, 2 * C, 0)
        N.check(_x3p_map(x, amap, 3 * C, [0, W * C, 2 * W * C], pp, Cout, y, Cout, None, 0, M,
                         None if bias is None else bias.detach()), "s2t_gemm_x3p_map(fwd)")
        ctx.save_for_backward(x, weight)
        ctx.params = (weight, bias)
        ctx.dims = (B, H, W, C, Ho, Wo, Cout)
        return y.view(B, Ho, Wo, Cout)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        B, H, W, C, Ho, Wo, Cout = ctx.dims
        g = dy.contiguous().float()
        has_bias = ctx.params[1] is not None
        dweight, db = side_param_grads(
            ctx.params, lambda: list(_conv3x3_wgrad_implicit(x, g.view(-1, Cout), 2, 2, has_bias)), keep=(x, g),
            allow=bool(_FRONT_SIDE & 2))
        dx = None
        if ctx.needs_input_grad[0]:
            gp = F.pad(g, (0, 0, 1, 1, 1, 1))                                  # zero border: every tap in range
            dx = torch.empty((B, H, W, C), dtype=torch.float32, device=g.device)
            wd = weight.detach()
            P2 = Wo + 2
            for ph in (0, 1):
                for pw in (0, 1):
                    taps = [(kh, kw) for kh in ((0, 2) if ph == 0 else (1,)) for kw in ((0, 2) if pw == 0 else (1,))]
                    # (Cin, taps * Cout): column t * Cout + cout = W[cout, cin, kh_t, kw_t]
                    bc = torch.cat([wd[:, :, kh, kw].t() for kh, kw in taps], dim=1).contiguous()
                    pp = planes.adhoc_pieces(bc, 0)
                    if pp is None:
                        raise RuntimeError("conv3x3 (map): class weight outside the piece kernel's rules")
                    Hc, Wc = (H - ph + 1) // 2, (W - pw + 1) // 2
                    amap = RowMap(Hc * Wc, Wc, (Ho + 2) * P2 * Cout, P2 * Cout, Cout, (P2 + 1) * Cout)
                    segoff = [((-1 if kh == 2 else 0) * P2 + (-1 if kw == 2 else 0)) * Cout for kh, kw in taps]
                    cmap = RowMap(Hc * Wc, Wc, H * W * C, 2 * W * C, 2 * C, (ph * W + pw) * C)
                    N.check(_x3p_map(gp, amap, Cout, segoff, pp, C, dx, C, cmap, dx.numel(), B * Hc * Wc, None),
                            "s2t_gemm_x3p_map(dgrad)")
        return dx, dweight, db


def conv3x3_s2_map_ok(x, weight, stride):
    """The implicit-operand GEMM serves this 3x3 convolution (channel-last x): stride 2 both ways, channel
    counts whose runs are whole 16-float stages."""
    return (x.is_cuda and tuple(stride) == (2, 2) and x.shape[-1] % 16 == 0 and weight.shape[0] % 16 == 0
            and x.shape[1] >= 3 and x.shape[2] >= 3 and x.numel() < (1 << 29) and weight.shape[1] == x.shape[-1])


def conv3x3_s2_map(x, weight, bias):
    return _Conv3x3S2Map.apply(x, weight, bias)


class _Conv3x3C1(torch.autograd.Function):
    """Conv2d(1, 8, 3, padding=(0, pw)) on (N,H,W,1) as a direct stencil (zip_front.hip): no
    padded copy, no im2col matrix, no 12-wide GEMM."""

    @staticmethod
    def forward(ctx, x, weight, bias, pw, grad_scale=None):
        _dev(x, weight)
        ctx.grad_scale = grad_scale
        x3 = x.reshape(x.shape[0], x.shape[1], x.shape[2]).contiguous().float()
        B, H, W = x3.shape
        CO = weight.shape[0]
        y = torch.empty((B, H - 2, W + 2 * pw - 2, CO), dtype=torch.float32, device=x.device)
        w = weight.contiguous().float()
        N.PROF[0] and N.profile_note("s2t_conv3x3_c1", 4.0 * (x3.numel() + y.numel()))
        N.check(N.lib().s2t_conv3x3_c1(0, N.fp(x3), N.fp(w), N.fp(bias), None, B, H, W, pw, CO,
                                       N.fp(y), None, None, None, N.stream()), "s2t_conv3x3_c1")
        ctx.save_for_backward(x3, w)
        ctx.cfg = (pw, bias is not None, x.shape)
        return y

    @staticmethod
    def backward(ctx, g):
        x3, w = ctx.saved_tensors
        pw, has_bias, xshape = ctx.cfg
        B, H, W = x3.shape
        CO = w.shape[0]
        g = g.contiguous().float()
        acc = torch.zeros(CO * 10, dtype=torch.float32, device=g.device)
        L = N.lib()
        N.PROF[0] and N.profile_note("s2t_conv3x3_c1", 4.0 * (x3.numel() + g.numel()))
        N.check(L.s2t_conv3x3_c1(1, N.fp(x3), None, None, N.fp(g), B, H, W, pw, CO, None, N.fp(acc),
                                 ctypes_off(acc, CO * 9), None, N.stream()), "s2t_conv3x3_c1(wgrad)")
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x3)
            N.PROF[0] and N.profile_note("s2t_conv3x3_c1", 4.0 * (g.numel() + dx.numel()))
            N.check(L.s2t_conv3x3_c1(2, None, N.fp(w), None, N.fp(g), B, H, W, pw, CO, None, None,
                                     None, N.fp(dx), N.stream()), "s2t_conv3x3_c1(dgrad)")
            dx = dx.view(xshape)
        if ctx.grad_scale is not None:
            # ScaleGrad(alpha) behind this conv (reference subsampling.py:217-218): the conv's
            # backward is linear in g, so alpha scales its (tiny) results, not the big map
            acc = acc * float(ctx.grad_scale)
            dx = None if dx is None else dx * float(ctx.grad_scale)
        return dx, acc[:CO * 9].view(w.shape), (acc[CO * 9:] if has_bias else None), None, None


class _Conv3x3S2(torch.autograd.Function):
    """Conv2d(8, 32, 3, stride 2) on (N,H,W,8) as direct kernels for the forward and the input
    gradient (zip_front.hip); the weight gradient is the TN GEMM over the patch matrix, which is
    built in backward only."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _dev(x, weight)
        x = x.contiguous().float()
        B, H, W, C = x.shape
        CO = weight.shape[0]
        w = weight.contiguous().float()
        y = torch.empty((B, (H - 3) // 2 + 1, (W - 3) // 2 + 1, CO), dtype=torch.float32,
                        device=x.device)
        N.PROF[0] and N.profile_note("s2t_conv3x3_s2", 4.0 * (x.numel() + y.numel()), 2.0 * y.numel() * 9 * C)
        N.check(N.lib().s2t_conv3x3_s2(0, N.fp(x), N.fp(w), N.fp(bias), None, B, H, W, C, CO,
                                       N.fp(y), None, N.stream()), "s2t_conv3x3_s2")
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        B, H, W, C = x.shape
        CO = w.shape[0]
        Ho, Wo = dy.shape[1], dy.shape[2]
        g = dy.contiguous().float()
        if _implicit_ok(x, CO, 2, 2):
            dweight, db = _conv3x3_wgrad_implicit(x, g, 2, 2, ctx.has_bias)
        else:
            s = x.stride()
            cols = x.as_strided((B, Ho, Wo, 3, 3 * C), (s[0], s[1] * 2, s[2] * 2, s[1], 1)) \
                .reshape(B * Ho * Wo, 9 * C)
            dwmat, db = linear_wgrad(g.view(B * Ho * Wo, CO), cols, ctx.has_bias)   # (CO, 9C)
            dweight = dwmat.view(CO, 3, 3, C).permute(0, 3, 1, 2)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            N.PROF[0] and N.profile_note("s2t_conv3x3_s2", 4.0 * (g.numel() + dx.numel()), 2.0 * g.numel() * 9 * C)
            N.check(N.lib().s2t_conv3x3_s2(2, None, N.fp(w), None, N.fp(g), B, H, W, C, CO, None,
                                           N.fp(dx), N.stream()), "s2t_conv3x3_s2(dgrad)")
        return dx, dweight, db


def conv3x3_direct_c1(x, weight, stride, pad_w):
    """True when conv3x3_nhwc runs the direct 1 -> 8 channel kernel (which can absorb a following
    ScaleGrad through `grad_scale`)."""
    return (x.shape[-1] == 1 and weight.shape[0] == 8 and tuple(stride) == (1, 1) and pad_w in (0, 1)
            and x.is_cuda)


def conv3x3_nhwc(x, weight, bias, stride=(1, 1), pad_w=0, grad_scale=None):
    """3x3 convolution on channel-last (N,H,W,Cin); pad_w = zero padding of the W axis (the
    reference's padding=(0, 1) of the first subsampling conv).  grad_scale: the gradients of this
    conv are multiplied by it (only with conv3x3_direct_c1)."""
    if conv3x3_direct_c1(x, weight, stride, pad_w):
        return _Conv3x3C1.apply(x, weight, bias, int(pad_w), grad_scale)
    if grad_scale is not None:
        raise ValueError("grad_scale needs the direct 1->8 channel kernel")
    if (x.shape[-1] == 8 and weight.shape[0] == 32 and tuple(stride) == (2, 2) and not pad_w
            and x.is_cuda and x.shape[0] <= 65535 and x.shape[1] <= 65535):
        return _Conv3x3S2.apply(x, weight, bias)
    if pad_w:
        x = F.pad(x, (0, 0, pad_w, pad_w))
    return _Conv3x3Nhwc.apply(x, weight, bias, int(stride[0]), int(stride[1]))


def _dwconv_wgrad(x, dy, w, has_bias, wshape):
    """[dW (weight's shape), db | None] of the depthwise conv (zip_front.hip), on the current stream."""
    Nn, H, W, C = x.shape
    KH, KW = w.shape[1], w.shape[2]
    L = N.lib()
    ws = torch.empty(L.s2t_dwconv2d_wgrad_workspace_floats(Nn, H, C, KH, KW), dtype=torch.float32,
                     device=x.device)
    dw = torch.empty_like(w)
    db = torch.empty(C, dtype=torch.float32, device=x.device) if has_bias else None
    N.PROF[0] and N.profile_note("s2t_dwconv2d_nhwc_wgrad", 8.0 * x.numel())
    N.check(L.s2t_dwconv2d_nhwc_wgrad(N.fp(x), N.fp(dy), Nn, H, W, C, KH, KW, N.fp(ws), N.fp(dw),
                                      N.fp(db), N.stream()), "s2t_dwconv2d_nhwc_wgrad")
    return [dw.view(wshape), db]


class _DwConv2dNhwc(torch.autograd.Function):
    """Depthwise 'same' conv on (N,H,W,C); HIP: zip_front.hip."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _dev(x, weight)
        x = x.contiguous().float()
        Nn, H, W, C = x.shape
        KH, KW = weight.shape[-2], weight.shape[-1]
        w = weight.reshape(C, KH, KW).contiguous()
        y = torch.empty_like(x)
        N.PROF[0] and N.profile_note("s2t_dwconv2d_nhwc_fwd", 8.0 * x.numel())
        N.check(N.lib().s2t_dwconv2d_nhwc_fwd(N.fp(x), N.fp(w), N.fp(bias), Nn, H, W, C, KH, KW, 0,
                                              N.fp(y), N.stream()), "s2t_dwconv2d_nhwc_fwd")
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.wshape = weight.shape
        ctx.params = (weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        Nn, H, W, C = x.shape
        KH, KW = w.shape[1], w.shape[2]
        dy = dy.contiguous().float()
        L = N.lib()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            N.PROF[0] and N.profile_note("s2t_dwconv2d_nhwc_fwd", 8.0 * dy.numel())
            N.check(L.s2t_dwconv2d_nhwc_fwd(N.fp(dy), N.fp(w), None, Nn, H, W, C, KH, KW, 1,
                                            N.fp(dx), N.stream()), "s2t_dwconv2d_nhwc_bwd_data")
        dw, db = side_param_grads(ctx.params, lambda: _dwconv_wgrad(x, dy, w, ctx.has_bias, ctx.wshape),
                                  keep=(x, dy), allow=bool(_FRONT_SIDE & 1))
        return dx, dw, db


def dwconv2d_nhwc(x, weight, bias):
    return _DwConv2dNhwc.apply(x, weight, bias)


class _DwConv2dTap(torch.autograd.Function):
    """(depthwise conv(x), x) -- the second output is x itself for the block's residual branch;
    that branch's gradient comes back here and is added inside the backward-data pass instead
    of by an autograd add over the (N,H,W,C) map."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        y = _DwConv2dNhwc.forward(ctx, x, weight, bias)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, g_pass):
        x, w = ctx.saved_tensors
        Nn, H, W, C = x.shape
        KH, KW = w.shape[1], w.shape[2]
        if g_pass is None or dy is None or (KH, KW) != (7, 7):
            dx, dw, db = _DwConv2dNhwc.backward(ctx, dy if dy is not None else torch.zeros_like(x))
            if g_pass is not None and dx is not None:
                dx = dx + g_pass
            return dx, dw, db
        dy = dy.contiguous().float()
        gp = g_pass.contiguous().float()
        L = N.lib()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            N.PROF[0] and N.profile_note("s2t_dwconv2d_nhwc_fwd_add", 12.0 * dy.numel())
            N.check(L.s2t_dwconv2d_nhwc_fwd_add(N.fp(dy), N.fp(w), None, N.fp(gp), Nn, H, W, C, KH, KW,
                                                1, N.fp(dx), N.stream()), "s2t_dwconv2d_nhwc_bwd_data")
        dw, db = side_param_grads(ctx.params, lambda: _dwconv_wgrad(x, dy, w, ctx.has_bias, ctx.wshape),
                                  keep=(x, dy), allow=bool(_FRONT_SIDE & 1))
        return dx, dw, db


def dwconv2d_nhwc_tap(x, weight, bias):
    """-> (depthwise conv(x), alias of x for the residual branch)."""
    return _DwConv2dTap.apply(x, weight, bias)
