"""Adam / AdamW on the flat parameter buffer.

Same update rule, hyper-parameters, defaults and state_dict layout as torch.optim.Adam / AdamW
(what the reference's OptimSetup hands out for `optimizer: type: "Adam" | "AdamW"`,
optimizer/optim_setup.py:364-385; every conformer YAML trains with AdamW), amsgrad / maximize /
capturable off.  torch runs ~8 foreach ops (~40 launches) per step plus 6 for the trainer's
grad-norm clip; with all parameters in one FlatStore (speech2text_amd.flat) a step is three HIP
launches (csrc/optim.hip): per-chunk sums of g^2, one workgroup that folds them into the clip
factor, one fused update that also zeroes the gradients.  `state[p]` holds views of the flat
moment buffers under torch's keys (`step`, `exp_avg`, `exp_avg_sq`), so checkpoints interchange.
Parameters that are not in a FlatStore on the GPU (host-logic tests) take torch's own step.
"""
import ctypes
import math

import torch
from torch.optim import Adam, AdamW

from speech2text_amd.flat import store_of


class _AdamGroup(ctypes.Structure):
    """Mirror of S2tAdamGroup (include/s2t_mi355.h)."""
    _fields_ = [("chunk_hi", ctypes.c_int), ("lr", ctypes.c_float), ("beta1", ctypes.c_float),
                ("beta2", ctypes.c_float), ("eps", ctypes.c_float),
                ("weight_decay", ctypes.c_float), ("bias_correction1", ctypes.c_float),
                ("sqrt_bias_correction2", ctypes.c_float), ("decoupled", ctypes.c_int)]


class _FlatMixin:
    fused_clip = True      # the trainer hands gradient_clip_val to `pre_clip` instead of clipping
    _decoupled = True

    def _flat_setup(self):
        self.pre_clip = None
        self.zero_grad_in_step = False
        # device flag (1 float) or None: != 0 turns this step into a no-op on parameters and state
        # (set by the trainer from the DP reducer's "step dropped on every rank" flag; no host sync)
        self.skip_flag = None
        self._flat = None

    def _flat_init(self):
        groups = [[p for p in g["params"] if p.requires_grad] for g in self.param_groups]
        allp = [p for ps in groups for p in ps]
        try:
            st = store_of(allp)
        except RuntimeError:
            st = None
        if st is None or not st.flat_p.is_cuda or len(groups) > 8:
            self._flat = False
            return
        for g in self.param_groups:
            if g.get("amsgrad") or g.get("maximize") or g.get("capturable") or \
                    g.get("differentiable") or torch.is_tensor(g["lr"]):
                self._flat = False
                return
        tb = st.tables()
        begin = tb["seg_chunk_begin"].tolist()
        his, prev = [], 0
        for ps in groups:
            try:
                lo, hi = st.range_of(ps)
            except RuntimeError:
                lo, hi = -1, -1
            if lo != prev:                              # groups must tile the store in order
                self._flat = False
                return
            his.append(begin[hi])
            prev = hi
        dev = st.flat_p.device
        self._flat = dict(store=st, chunk_hi=his, m=torch.zeros_like(st.flat_p),
                          v=torch.zeros_like(st.flat_p),
                          partial=torch.zeros(tb["nchunks"] * 3, device=dev),
                          coef=torch.ones(2, device=dev), step=0)
        f = self._flat
        # torch's per-parameter state, for the parameters the groups list (a store may hold
        # trainable parameters outside every group: they are only zeroed, and a state entry for
        # them would break Optimizer.state_dict()'s id mapping)
        grouped = {id(p) for ps in groups for p in ps}
        f["step_t"] = torch.tensor(0.0)
        for p, o, n in zip(st.params, st.offsets, st.lengths):
            if id(p) in grouped:
                self.state[p] = {"step": f["step_t"], "exp_avg": f["m"][o:o + n].view(p.shape),
                                 "exp_avg_sq": f["v"][o:o + n].view(p.shape)}

    @torch.no_grad()
    def step(self, closure=None):
        if self._flat is None:
            self._flat_init()
        if self._flat is False:
            if self.skip_flag is not None and float(self.skip_flag) != 0.0:
                if self.zero_grad_in_step:   # dropped step (ddp.py), host form: CPU tensors only
                    self.zero_grad(set_to_none=False)
                return None
            if self.pre_clip:
                params = [p for g in self.param_groups for p in g["params"] if p.grad is not None]
                torch.nn.utils.clip_grad_norm_(params, self.pre_clip)
            out = super().step(closure)
            if self.zero_grad_in_step:
                self.zero_grad(set_to_none=False)
            return out
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        from speech2text_amd import _native as N
        f = self._flat
        st = f["store"]
        st.check_views()
        tb = st.tables()
        L, stream = N.lib(), N.stream()
        nbytes = 4.0 * st.numel
        N.profile_note("s2t_seg_stats", 2 * nbytes)
        N.check(L.s2t_seg_stats(N.fp(st.flat_p), N.fp(st.flat_g), N.ip(tb["chunk_off"]),
                                N.ip(tb["chunk_len"]), tb["nchunks"], N.fp(f["partial"]), stream),
                "s2t_seg_stats")
        N.check(L.s2t_clip_coef(N.fp(f["partial"]), tb["nchunks"], float(self.pre_clip or 0.0),
                                N.fp(f["coef"]), stream), "s2t_clip_coef")
        f["step"] += 1
        k = f["step"]
        arr = (_AdamGroup * len(self.param_groups))()
        for q, g, hi in zip(arr, self.param_groups, f["chunk_hi"]):
            b1, b2 = g["betas"]
            q.chunk_hi, q.lr, q.beta1, q.beta2, q.eps = hi, float(g["lr"]), b1, b2, float(g["eps"])
            q.weight_decay = float(g["weight_decay"])
            q.bias_correction1 = 1.0 - b1 ** k
            q.sqrt_bias_correction2 = math.sqrt(1.0 - b2 ** k)
            q.decoupled = int(self._decoupled)
        N.profile_note("s2t_adam_apply", 8 * nbytes)
        N.check(L.s2t_adam_apply(N.fp(st.flat_p), N.fp(st.flat_g), N.fp(f["m"]), N.fp(f["v"]),
                                 N.ip(tb["chunk_off"]), N.ip(tb["chunk_len"]), tb["nchunks"],
                                 len(self.param_groups), ctypes.cast(arr, ctypes.c_void_p),
                                 N.fp(f["coef"]), int(self.zero_grad_in_step), N.fp(self.skip_flag),
                                 stream),
                "s2t_adam_apply")
        st.epoch += 1                      # parameters rewritten: weight pieces are stale
        f["step_t"].fill_(float(k))        # ONE shared host scalar: every state's "step" is this tensor
        return loss

    def load_state_dict(self, sd):
        """torch's loader replaces the per-parameter tensors; copy them back into the flat moment
        buffers so the fused step continues from the loaded state."""
        if self._flat is None:
            self._flat_init()
        if self._flat is False:
            return super().load_state_dict(sd)
        f = self._flat
        st = f["store"]
        super().load_state_dict(sd)
        k = 0
        for p, o, n in zip(st.params, st.offsets, st.lengths):
            s = self.state.get(p)
            if not s:
                continue
            f["m"][o:o + n].copy_(s["exp_avg"].reshape(-1))
            f["v"][o:o + n].copy_(s["exp_avg_sq"].reshape(-1))
            k = max(k, int(float(s["step"])))
            s["exp_avg"] = f["m"][o:o + n].view(p.shape)
            s["exp_avg_sq"] = f["v"][o:o + n].view(p.shape)
            s["step"] = f["step_t"]
        f["step"] = k
        f["step_t"].fill_(float(k))


class FlatAdamW(_FlatMixin, AdamW):
    _decoupled = True

    def __init__(self, params, *args, **kwargs):
        super().__init__(params, *args, **kwargs)
        self._flat_setup()


class FlatAdam(_FlatMixin, Adam):
    _decoupled = False

    def __init__(self, params, *args, **kwargs):
        super().__init__(params, *args, **kwargs)
        self._flat_setup()
