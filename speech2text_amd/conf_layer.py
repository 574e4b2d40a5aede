"""One autograd node per conformer layer (training hot path of C2 / C4 / C5).

The module-by-module form of the layer (model/encoder/conformer.py: FFN(0.5) -> MHSA -> conv
module -> FFN(0.5) -> LayerNorm, torchaudio.models.Conformer as called at reference
model/encoder/conformer.py:170-178,193) costs autograd ~40 nodes per layer and leaves the
norms, activations and residual adds to generic elementwise kernels.  Here the layer is one
forward and one hand-scheduled backward over (T*B, D) row blocks:

  * the residual sums ride in a GEMM epilogue (`x + module(x)`) or in the next LayerNorm's
    kernel (`x + 0.5 * ffn(x)`): no stand-alone add passes;
  * LayerNorm backward adds the residual branch's gradient and accumulates d gamma / d beta
    straight into the flat gradient buffer; BatchNorm + SiLU is a statistics pass + one fused
    pass each way; the attention core is the flash-style MFMA kernel (csrc/conf_attn.hip);
  * the 8 weight / bias gradients of the layer's Linears go out as ONE grouped TN launch on the
    side stream at the end of the layer's backward (the 0.5 of the feed-forward residuals rides
    in the problem's `alpha`);
  * forward / data-gradient GEMMs are hipBLASLt with bias (+ residual) in the epilogue.

  * dropout (the YAMLs ship 0.1): every nn.Dropout site is a stateless hash of (seed, element)
    regenerated in backward -- after the feed-forward SiLU it rides in the activation kernel, on
    the attention probabilities inside the MHSA kernel, after a module's last Linear it is one
    `x + alpha * drop(y)` pass instead of the GEMM's residual epilogue.

Cases the executor does not cover (GroupNorm, convolution_first, parameters outside a FlatStore,
evaluation) run the module path, which uses the same kernels one op at a time.
"""
import os

import torch

from . import _native as N
from . import conf_kernels as ck
from . import flat
from . import zip_kernels as zk

_F32 = torch.float32
ENABLED = os.environ.get("S2T_CONF_EXEC", "1") == "1"
CALLS = [0]          # layer calls served by the executor (tests assert the path really ran)


def _static_ok(layer):
    ok = layer.__dict__.get("_cl_static")
    if ok is None:
        mha = layer.self_attn
        D = mha.embed_dim
        H = mha.num_heads
        cm = layer.conv_module
        pw1, _, dw, norm, _, pw2, _ = cm.sequential
        ok = (D % 4 == 0 and D <= 1024 and D % H == 0 and (D // H) in (16, 32, 64)
              and isinstance(norm, torch.nn.BatchNorm1d) and not layer.convolution_first
              and mha.in_proj_weight is not None and mha.in_proj_bias is not None
              and mha.bias_k is None and not mha.add_zero_attn
              and layer.ffn1.sequential[1].out_features % 4 == 0
              and dw.kernel_size[0] in (7, 15, 31) and pw1.bias is not None
              and pw2.bias is not None and dw.bias is not None)
        layer.__dict__["_cl_static"] = ok
    return ok


def eligible(layer, x):
    if not (ENABLED and layer.training and torch.is_grad_enabled() and x.is_cuda
            and x.dtype == _F32 and x.dim() == 3 and x.shape[0] >= 4):
        return False
    if not _static_ok(layer):
        return False
    for m in (layer.ffn1.sequential[3], layer.ffn1.sequential[5], layer.ffn2.sequential[3],
              layer.ffn2.sequential[5], layer.self_attn_dropout, layer.conv_module.sequential[6]):
        if not 0.0 <= float(m.p) < 1.0:
            return False
    if not 0.0 <= float(layer.self_attn.dropout) < 1.0:
        return False
    p = layer.final_layer_norm.weight
    return flat.owned(p) and p.grad is not None


def run(layer, x, lengths):
    """x (T,B,D) time-major, lengths (B,) int64 valid frames -> layer output (T,B,D)."""
    CALLS[0] += 1
    return _LayerFn.apply(x, layer, lengths)


class _Saved:
    pass


def _seed(p):
    return ck.draw_seed() if p > 0.0 else 0


def _ffn_fwd(ffn, n, sv):
    """Linear -> SiLU -> Dropout -> Linear (the trailing Dropout is applied where the residual sum
    is formed).  sv keeps (p, seed) of both dropout sites."""
    l1, l2 = ffn.sequential[1], ffn.sequential[4]
    sv.p_act, sv.p_out = float(ffn.sequential[3].p), float(ffn.sequential[5].p)
    sv.seed_act, sv.seed_out = _seed(sv.p_act), _seed(sv.p_out)
    h = zk.lt_matmul(0, n, l1.weight, l1.bias)
    a = ck.silu_fwd(h, sv.p_act, sv.seed_act)   # kept for the weight gradient of the second Linear
    y = zk.lt_matmul(0, a, l2.weight, l2.bias)
    return h, a, y


def _ffn_sum_ln(x_in, y, sv, ln):
    """(x_in + 0.5 * drop(y), LN of it, LN stats): without dropout the sum is formed by the
    LayerNorm kernel itself."""
    if sv.p_out > 0.0:
        xs = ck.dropout_add(x_in, y, 0.5, sv.p_out, sv.seed_out)
        _, n, st = ck.ln_fwd(xs, None, 0.0, ln.weight, ln.bias, ln.eps)
        return xs, n, st
    return ck.ln_fwd(x_in, y, 0.5, ln.weight, ln.bias, ln.eps)


def _ln_bwd(ln, lnp, x_in, stats, dn, resid):
    """LayerNorm backward (+ the residual branch's gradient); the parameter gradients are folded
    at the end of the layer's backward (one launch for its five LayerNorms)."""
    part = []
    dx = ck.ln_bwd(x_in, stats, ln.weight, dn, resid, part)
    lnp.append(part[0] + (ln.weight.grad, ln.bias.grad))
    lnp.params.extend((ln.weight, ln.bias))
    return dx


class _LnPend(list):
    def __init__(self):
        super().__init__()
        self.params = []


def _ffn_bwd(ffn, pend, lnp, n, h, a, g, ln, x_in, stats, sv):
    """g = gradient w.r.t. the sum x_in + 0.5 * drop(ffn(LN(x_in))) -> gradient w.r.t. x_in."""
    l1, l2 = ffn.sequential[1], ffn.sequential[4]
    gy = ck.dropout_add(None, g, 1.0, sv.p_out, sv.seed_out) if sv.p_out > 0.0 else g
    pend.append((l2.weight, l2.bias, gy, a, 0.5))
    dh = ck.silu_bwd(h, zk.lt_matmul(1, gy, l2.weight), 0.5, True, sv.p_act, sv.seed_act)
    pend.append((l1.weight, l1.bias, dh, n))
    dn = zk.lt_matmul(1, dh, l1.weight)
    return _ln_bwd(ln, lnp, x_in, stats, dn, g)


def _resid_drop(x_in, y_of, p, seed):
    """x_in + drop(module output): y_of(resid) runs the module's last GEMM, with the residual in
    its epilogue when there is no dropout."""
    if p > 0.0:
        return ck.dropout_add(x_in, y_of(None), 1.0, p, seed)
    return y_of(x_in)


class _LayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, layer, lengths):
        T, B, D = x.shape
        R = T * B
        x0 = x.contiguous().view(R, D)
        if x0.data_ptr() % 16:
            x0 = x0.clone()
        s = _Saved()
        s.dims = (T, B, D)
        s.lens = None if lengths is None else lengths.to(device=x.device,
                                                         dtype=torch.int64).contiguous()
        mha = layer.self_attn
        H = mha.num_heads
        cm = layer.conv_module
        pw1, _, dw, bn, _, pw2, _ = cm.sequential

        # FFN1: x1 = x0 + 0.5 * ffn1(x0)  (the sum is formed by the next LayerNorm's kernel)
        ln1 = layer.ffn1.sequential[0]
        _, s.n1, s.st1 = ck.ln_fwd(x0, None, 0.0, ln1.weight, ln1.bias, ln1.eps)
        s.f1 = _Saved()
        s.h1, s.a1, y1 = _ffn_fwd(layer.ffn1, s.n1, s.f1)
        s.x1, s.n2, s.st2 = _ffn_sum_ln(x0, y1, s.f1, layer.self_attn_layer_norm)
        # MHSA: x2 = x1 + drop(out_proj(attn(in_proj(LN(x1)))))
        s.p_attn, s.p_sa = float(mha.dropout), float(layer.self_attn_dropout.p)
        s.seed_attn, s.seed_sa = _seed(s.p_attn), _seed(s.p_sa)
        s.qkv = zk.lt_matmul(0, s.n2, mha.in_proj_weight, mha.in_proj_bias)
        s.o, s.lse = ck.mhsa_fwd(s.qkv, s.lens, T, B, H, s.p_attn, s.seed_attn)
        s.x2 = _resid_drop(s.x1, lambda r: zk.lt_matmul(0, s.o, mha.out_proj.weight,
                                                        mha.out_proj.bias, r), s.p_sa, s.seed_sa)
        # conv module: x3 = x2 + pw2(SiLU(BN(dwconv(GLU(pw1(LN(x2)))))))
        lnc = cm.layer_norm
        _, s.n3, s.st3 = ck.ln_fwd(s.x2, None, 0.0, lnc.weight, lnc.bias, lnc.eps)
        s.u = zk.lt_matmul(0, s.n3, pw1.weight.view(2 * D, D), pw1.bias)
        s.cp = zk.conv_params(dw, T, -1)
        s.c = zk.zipconv_forward(s.u.view(T, B, 2 * D), D, None, *s.cp).view(R, D)
        s.sb, s.bn_mean, s.bn_rstd = ck.bn_silu_fwd(s.c, bn)
        s.p_cv = float(cm.sequential[6].p)
        s.seed_cv = _seed(s.p_cv)
        s.x3 = _resid_drop(s.x2, lambda r: zk.lt_matmul(0, s.sb, pw2.weight.view(D, D), pw2.bias, r),
                           s.p_cv, s.seed_cv)
        # FFN2 + final LayerNorm: out = LN(x3 + 0.5 * ffn2(x3))
        ln2 = layer.ffn2.sequential[0]
        _, s.n4, s.st4 = ck.ln_fwd(s.x3, None, 0.0, ln2.weight, ln2.bias, ln2.eps)
        s.f2 = _Saved()
        s.h2, s.a2, y2 = _ffn_fwd(layer.ffn2, s.n4, s.f2)
        s.x4, out, s.st5 = _ffn_sum_ln(s.x3, y2, s.f2, layer.final_layer_norm)
        s.x0 = x0
        ctx.s, ctx.layer = s, layer
        return out.view(T, B, D)

    @staticmethod
    def backward(ctx, g):
        s, layer = ctx.s, ctx.layer
        ctx.s = None
        T, B, D = s.dims
        R = T * B
        mha = layer.self_attn
        H = mha.num_heads
        cm = layer.conv_module
        pw1, _, dw, bn, _, pw2, _ = cm.sequential
        g = g.contiguous().view(R, D)
        if g.dtype != _F32:
            g = g.float()
        if g.data_ptr() % 16:
            g = g.clone()
        pend = []
        lnp = _LnPend()

        g4 = _ln_bwd(layer.final_layer_norm, lnp, s.x4, s.st5, g, None)
        g3 = _ffn_bwd(layer.ffn2, pend, lnp, s.n4, s.h2, s.a2, g4, layer.ffn2.sequential[0], s.x3,
                      s.st4, s.f2)

        # conv module
        gy = ck.dropout_add(None, g3, 1.0, s.p_cv, s.seed_cv) if s.p_cv > 0.0 else g3
        pend.append((pw2.weight, pw2.bias, gy, s.sb))
        ds = zk.lt_matmul(1, gy, pw2.weight.view(D, D))
        dc = ck.bn_silu_bwd(s.c, ds, s.bn_mean, s.bn_rstd, bn.weight, bn.bias, bn.weight.grad,
                            bn.bias.grad)
        flat.grad_written(bn.weight)
        flat.grad_written(bn.bias)
        chunk, K, wc, bc, wk, bk, scale = s.cp
        du = zk.zipconv_backward(s.u.view(T, B, 2 * D), D, None, chunk, K, None, wk, bk, None,
                                 dc.view(T, B, D), (None, None, dw.weight.grad, dw.bias.grad, None))
        flat.grad_written(dw.weight)
        flat.grad_written(dw.bias)
        du = du.view(R, 2 * D)
        pend.append((pw1.weight, pw1.bias, du, s.n3))
        dn3 = zk.lt_matmul(1, du, pw1.weight.view(2 * D, D))
        g2 = _ln_bwd(cm.layer_norm, lnp, s.x2, s.st3, dn3, g3)

        # MHSA
        gy = ck.dropout_add(None, g2, 1.0, s.p_sa, s.seed_sa) if s.p_sa > 0.0 else g2
        pend.append((mha.out_proj.weight, mha.out_proj.bias, gy, s.o))
        do = zk.lt_matmul(1, gy, mha.out_proj.weight)
        dqkv = ck.mhsa_bwd(s.qkv, s.lens, T, B, H, s.o, do, s.lse, s.p_attn, s.seed_attn)
        pend.append((mha.in_proj_weight, mha.in_proj_bias, dqkv, s.n2))
        dn2 = zk.lt_matmul(1, dqkv, mha.in_proj_weight)
        g1 = _ln_bwd(layer.self_attn_layer_norm, lnp, s.x1, s.st2, dn2, g2)

        g0 = _ffn_bwd(layer.ffn1, pend, lnp, s.n1, s.h1, s.a1, g1, layer.ffn1.sequential[0], s.x0,
                      s.st1, s.f1)
        ck.ln_param_grad(lnp, D)
        for p in lnp.params:
            flat.grad_written(p)
        zk.wgrad_group(pend)
        return g0.view(T, B, D), None, None
