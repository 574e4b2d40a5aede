"""ScaledAdam on flat buffers.

Same update rule, hyper-parameters and defaults as the reference's
optimizer/scaled_adam.py:112-736 (per-tensor RMS-scaled Adam step, learned tensor scale
every `size_update_period` steps, median-based gradient clipping, scalar rule for 1-element
tensors).  The reference stacks same-shaped tensors into batches every step (:30-109); here
each param group lives in one flat buffer (speech2text_amd.flat.FlatStore) and per-tensor
statistics are segmented reductions, so a step is ~25 launches regardless of tensor count
and never synchronises with the host except when the clipping threshold is re-estimated.
"""
import torch
from torch.optim import Optimizer

from speech2text_amd.flat import get_store


class ScaledAdam(Optimizer):
    def __init__(self, params, lr=3e-02, clipping_scale=None, betas=(0.9, 0.98),
                 scalar_lr_scale=0.1, eps=1.0e-08, param_min_rms=1.0e-05, param_max_rms=3.0,
                 scalar_max=10.0, size_update_period=4, clipping_update_period=100):
        params = list(params)
        if len(params) == 0:
            raise ValueError("optimizer got an empty parameter list")
        if not isinstance(params[0], dict):
            params = [p[1] if isinstance(p, tuple) else p for p in params]
        else:
            for g in params:
                if "named_params" in g:
                    g["params"] = [x[1] for x in g.pop("named_params")]
        defaults = dict(lr=lr, clipping_scale=clipping_scale, betas=betas,
                        scalar_lr_scale=scalar_lr_scale, eps=eps, param_min_rms=param_min_rms,
                        param_max_rms=param_max_rms, scalar_max=scalar_max,
                        size_update_period=size_update_period,
                        clipping_update_period=clipping_update_period)
        super().__init__(params, defaults)
        self._gstate = [None] * len(self.param_groups)

    # ------------------------------------------------------------------
    def _init_group(self, gi, group):
        ps = [p for p in group["params"] if p.requires_grad]
        st = get_store(ps)
        dev = st.flat_p.device
        n = len(st.lengths)
        P = group["size_update_period"]
        lens = st.seg_lengths.to(torch.float32)
        s = dict(store=st, step=0, lens=lens, scalar=(st.seg_lengths == 1),
                 has_scalar=any(n == 1 for n in st.lengths),
                 delta=torch.zeros(st.numel, device=dev),
                 exp_avg_sq=torch.zeros(st.numel, device=dev),
                 scale_exp_avg_sq=torch.zeros(n, device=dev),
                 scale_grads=torch.zeros(P, n, device=dev),
                 model_norms=torch.zeros(group["clipping_update_period"], device=dev),
                 threshold=None, num_clipped=torch.zeros((), device=dev))
        p = st.p()
        s["param_rms"] = (st.seg_sum(p * p) / lens).sqrt()
        self._gstate[gi] = s
        return s

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            s = self._gstate[gi] or self._init_group(gi, group)
            self._step_group(group, s)
        return loss

    def _clip(self, group, s, g, gsq_seg):
        cs = group["clipping_scale"]
        k = s["step"]
        if cs is None or k == 0:
            return
        period = group["clipping_update_period"]
        slr = group["scalar_lr_scale"]
        w = torch.where(s["scalar"], torch.full_like(s["param_rms"], slr * slr),
                        s["param_rms"] * s["param_rms"])
        tot_norm = (gsq_seg * w).sum().sqrt()
        s["model_norms"][k % period] = tot_norm
        irregular = [i for i in (10, 20, 40) if i < period]
        if k % period == 0 or k in irregular:
            sorted_norms = s["model_norms"].sort()[0]
            if k in irregular:
                sorted_norms = sorted_norms[-k:]
            num = sorted_norms.numel()
            median = sorted_norms[min(num - 1, (num // 4) * 2)]
            if not bool(torch.isfinite(median)):          # rare, host sync only here
                raise RuntimeError("Too many grads were not finite")
            thr = cs * median
            if k in irregular:
                thr = thr * 2.0
            s["threshold"] = thr
            s["num_clipped"].zero_()
        if s["threshold"] is None:
            return
        ans = torch.clamp(s["threshold"] / (tot_norm + 1.0e-20), max=1.0)
        ans = torch.nan_to_num(ans, nan=0.0)
        s["num_clipped"] += (ans < 1.0)
        g.mul_(ans)
        # reference zeroes the grads when the factor is 0 (inf/nan grads): 0 * inf = nan
        torch.nan_to_num_(g, nan=0.0, posinf=0.0, neginf=0.0)

    def _step_group(self, group, s):
        st = s["store"]
        st.check_views()
        p, g = st.p(), st.g()
        k = s["step"]
        lr = group["lr"]
        beta1, beta2 = group["betas"]
        eps = group["eps"]
        P = group["size_update_period"]
        slr = group["scalar_lr_scale"]
        seg = st.seg_ids
        scalar = s["scalar"]
        delta, eas = s["delta"], s["exp_avg_sq"]
        gsq = g * g
        self._clip(group, s, g, st.seg_sum(gsq))
        if group["clipping_scale"] is not None and k > 0 and s["threshold"] is not None:
            gsq = g * g
        delta.mul_(beta1)
        # ---- learned tensor scale (non-scalar tensors)
        s["scale_grads"][k % P] = st.seg_sum(p * g)
        if k % P == P - 1:
            s["param_rms"] = (st.seg_sum(p * p) / s["lens"]).sqrt()
            if k > 0:
                rms = s["param_rms"]
                beta2c = beta2 ** P
                sg = s["scale_grads"]
                s["scale_exp_avg_sq"].mul_(beta2c).add_((sg * sg).mean(dim=0), alpha=1 - beta2c)
                size_step = (k + 1) // P
                bc2 = 1 - beta2c ** size_step
                denom = s["scale_exp_avg_sq"].sqrt() + eps
                scale_step = -(lr * slr) * (bc2 ** 0.5) * sg.sum(dim=0) / denom
                scale_step = scale_step.masked_fill(rms < group["param_min_rms"], 0.0)
                scale_step = torch.minimum(scale_step, (group["param_max_rms"] - rms) / rms)
                scale_step = scale_step.masked_fill(scalar, 0.0)
                delta.add_(p * scale_step[seg], alpha=(1 - beta1))
        # ---- Adam-like step, scaled by the tensor rms (or the scalar rule)
        eas.mul_(beta2).add_(gsq, alpha=1 - beta2)
        bc2 = 1 - beta2 ** (k + 1)
        bc_vec = torch.where(scalar, torch.full_like(s["lens"], bc2),
                             torch.full_like(s["lens"], bc2 if bc2 < 0.99 else 1.0))
        coef = torch.where(scalar, torch.full_like(s["lens"], -lr * slr * (1 - beta1)),
                           -lr * (1 - beta1) * s["param_rms"].clamp(min=group["param_min_rms"]))
        denom = (eas / bc_vec[seg]).sqrt_().add_(eps)
        delta.add_(g / denom * coef[seg])
        # scalar parameters are clamped before the update, as in the reference
        if s["has_scalar"]:
            lim = torch.where(scalar, torch.full_like(s["lens"], group["scalar_max"]),
                              torch.full_like(s["lens"], float("inf")))[seg]
            torch.minimum(p, lim, out=p)
            torch.maximum(p, -lim, out=p)
        p.add_(delta)
        s["step"] = k + 1

    def zero_grad(self, set_to_none: bool = False):
        done = False
        for s in self._gstate:
            if s is not None:
                s["store"].zero_grad()
                done = True
        if not done:
            super().zero_grad(set_to_none=False)
