"""ScaledAdam on a flat buffer.

Same update rule, hyper-parameters and defaults as the reference's
optimizer/scaled_adam.py:112-736 (per-tensor RMS-scaled Adam step, learned tensor scale
every `size_update_period` steps, median-based gradient clipping, scalar rule for 1-element
tensors).  The reference stacks same-shaped tensors into batches every step (:30-109) and runs
~20 torch ops per batch; here all parameters live in one flat buffer
(speech2text_amd.flat.FlatStore, param groups = contiguous tensor ranges) and a step is three
HIP launches (csrc/optim.hip): per-chunk sums, one coefficient workgroup per param group, one
fused update that also applies the trainer's grad-norm clip and zeroes the gradients.  The host
never waits for the device except when the clipping threshold is re-estimated (every
`clipping_update_period` steps), to raise on a non-finite median as the reference does.

CPU tensors (host-logic tests only) run the same arithmetic as torch ops on the same buffers.
"""
import torch
from torch.optim import Optimizer

from speech2text_amd.flat import FlatStore, store_of


class ScaledAdam(Optimizer):
    fused_clip = True      # the trainer hands gradient_clip_val to `pre_clip` instead of clipping

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
        self._gstate = None
        self.store = None
        self.pre_clip = None          # trainer's gradient_clip_val (norm), applied inside step()
        self.zero_grad_in_step = False
        # device flag (1 float) or None: != 0 turns this step into a no-op on parameters and state
        # (set by the trainer from the DP reducer's "step dropped on every rank" flag; no host sync)
        self.skip_flag = None
        self.last_clip = None

    # ------------------------------------------------------------------ state
    def _init(self):
        groups = [[p for p in g["params"] if p.requires_grad] for g in self.param_groups]
        allp = [p for ps in groups for p in ps]
        st = store_of(allp)
        if st is None:
            st = FlatStore(allp)
        self.store = st
        dev = st.flat_p.device
        self._delta = torch.zeros_like(st.flat_p)
        self._eas = torch.zeros_like(st.flat_p)
        tb = st.tables()
        self._partial = torch.zeros(tb["nchunks"] * 3, device=dev)
        self._segstat = torch.zeros(tb["nseg"] * 3, device=dev)
        self._segc = None
        self._gstate = []
        for ps, group in zip(groups, self.param_groups):
            lo, hi = st.range_of(ps)
            n = hi - lo
            P = group["size_update_period"]
            lens = st.seg_lengths[lo:hi].to(torch.float32)
            f_lo, f_hi = st.offsets[lo], (st.offsets[hi] if hi < len(st.offsets) else st.numel)
            s = dict(lo=lo, hi=hi, f_lo=f_lo, f_hi=f_hi, step=0, lens=lens,
                     scale_exp_avg_sq=torch.zeros(n, device=dev),
                     scale_grads=torch.zeros(P, n, device=dev),
                     model_norms=torch.zeros(group["clipping_update_period"], device=dev),
                     fstate=torch.zeros(3, device=dev),
                     istate=torch.zeros(3, dtype=torch.int32, device=dev))
            # segment ids of the group's flat range (padding belongs to the tensor it follows)
            padded = [(st.offsets[i + 1] if i + 1 < len(st.offsets) else st.numel) - st.offsets[i]
                      for i in range(lo, hi)]
            s["padded"] = torch.tensor(padded, device=dev)
            s["seg"] = torch.repeat_interleave(torch.arange(n, device=dev), s["padded"])
            p = st.flat_p[f_lo:f_hi]
            s["param_rms"] = (self._seg_sum(s, p * p) / lens).sqrt()
            self._gstate.append(s)

    @staticmethod
    def _seg_sum(s, x):
        # segment_reduce is deterministic (index_add_ uses atomics on the GPU: replicas of a
        # data-parallel job would drift apart in the last bit)
        return torch.segment_reduce(x, "sum", lengths=s["padded"], unsafe=True)

    # ------------------------------------------------------------------ step
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._gstate is None:
            self._init()
        st = self.store
        st.check_views()
        if st.flat_p.is_cuda:
            self._step_hip()
        elif self.skip_flag is not None and float(self.skip_flag) != 0.0:
            # dropped step (ddp.py), host form (CPU tensors only).  As on the GPU path the step
            # count (bias correction, LR schedule position) advances, parameters and moments stay,
            # and the clipping window repeats the previous norm.
            for s in self._gstate:
                k = s["step"]
                mn = s.get("model_norms")
                if mn is not None and k > 0:
                    P = mn.numel()
                    mn[k % P] = mn[(k - 1) % P]
                s["step"] = k + 1
            if self.zero_grad_in_step:
                st.flat_g.zero_()
        else:
            self._step_torch()
        return loss

    def _host_scalars(self, group, s):
        k = s["step"]
        beta1, beta2 = group["betas"]
        P = group["size_update_period"]
        beta2c = beta2 ** P
        return dict(k=k, beta1=beta1, beta2=beta2, P=P, beta2c=beta2c, bc2=1 - beta2 ** (k + 1),
                    bc2_size=1 - beta2c ** ((k + 1) // P))

    def _step_hip(self):
        from speech2text_amd import _native as N
        st = self.store
        tb = st.tables()
        L = N.lib()
        stream = N.stream()
        if self._segc is None:
            self._segc = torch.zeros(tb["nseg"] * L.s2t_optim_segc_floats(), device=st.flat_p.device)
            assert L.s2t_optim_chunk_elems() == 8192
        nbytes = 4.0 * st.numel
        N.profile_note("s2t_seg_stats", 2 * nbytes)
        N.check(L.s2t_seg_stats(N.fp(st.flat_p), N.fp(st.flat_g), N.ip(tb["chunk_off"]),
                                N.ip(tb["chunk_len"]), tb["nchunks"], N.fp(self._partial), stream),
                "s2t_seg_stats")
        rare = []
        for group, s in zip(self.param_groups, self._gstate):
            h = self._host_scalars(group, s)
            k = h["k"]
            cs = group["clipping_scale"]
            period = group["clipping_update_period"]
            N.profile_note("s2t_scaled_adam_coef", 4.0 * (self._partial.numel() + self._segstat.numel()
                                                          + self._segc.numel()))
            N.check(L.s2t_scaled_adam_coef(
                N.fp(self._partial), N.ip(tb["seg_chunk_begin"]), N.ip(tb["seg_len"]),
                tb["nchunks"], s["lo"], s["hi"], float(group["lr"]), h["beta1"], h["beta2"],
                float(group["eps"]), float(group["scalar_lr_scale"]), float(group["param_min_rms"]),
                float(group["param_max_rms"]), float(group["scalar_max"]),
                float(self.pre_clip or 0.0), float(cs or 0.0), k, h["P"], period, h["bc2"],
                h["bc2_size"], h["beta2c"], N.fp(s["param_rms"]), N.fp(s["scale_exp_avg_sq"]),
                N.fp(s["scale_grads"]), N.fp(s["model_norms"]), N.fp(s["fstate"]),
                N.ip(s["istate"]), N.fp(self._segstat), N.fp(self._segc), N.fp(self.skip_flag),
                stream),
                "s2t_scaled_adam_coef")
            if cs is not None and k > 0 and (k % period == 0 or (k in (10, 20, 40) and k < period)):
                rare.append(s)
            s["step"] = k + 1
        for s in rare:                                   # the only host sync (every `period` steps),
            if int(s["istate"][2]) != 0:                 # BEFORE the update touches the parameters
                for s2 in self._gstate:
                    s2["step"] -= 1
                raise RuntimeError("Too many grads were not finite")
        N.profile_note("s2t_scaled_adam_apply", 8 * nbytes)
        N.check(L.s2t_scaled_adam_apply(N.fp(st.flat_p), N.fp(st.flat_g), N.fp(self._delta),
                                        N.fp(self._eas), N.ip(tb["chunk_off"]),
                                        N.ip(tb["chunk_len"]), N.ip(tb["chunk_seg"]), tb["nchunks"],
                                        N.fp(self._segc), int(self.zero_grad_in_step), stream),
                "s2t_scaled_adam_apply")
        st.epoch += 1                      # parameters rewritten: weight pieces are stale

    # ---- the same arithmetic as torch ops (CPU tensors: host-logic tests)
    def _step_torch(self):
        st = self.store
        c = None
        if self.pre_clip:
            c = torch.clamp(self.pre_clip / (st.g().norm() + 1.0e-6), max=1.0)
        for group, s in zip(self.param_groups, self._gstate):
            h = self._host_scalars(group, s)
            k, beta1, beta2, P = h["k"], h["beta1"], h["beta2"], h["P"]
            p = st.flat_p[s["f_lo"]:s["f_hi"]]
            g = st.flat_g[s["f_lo"]:s["f_hi"]]
            delta = self._delta[s["f_lo"]:s["f_hi"]]
            eas = self._eas[s["f_lo"]:s["f_hi"]]
            seg, lens = s["seg"], s["lens"]
            scalar = lens == 1
            lr, slr, eps = group["lr"], group["scalar_lr_scale"], group["eps"]
            if c is not None:
                g.mul_(c)
            cs, period = group["clipping_scale"], group["clipping_update_period"]
            if cs is not None and k > 0:
                gsq_seg = self._seg_sum(s, g * g)
                w = torch.where(scalar, torch.full_like(lens, slr * slr),
                                s["param_rms"] * s["param_rms"])
                tot_norm = (gsq_seg * w).sum().sqrt()
                s["model_norms"][k % period] = tot_norm
                irregular = k in (10, 20, 40) and k < period
                if k % period == 0 or irregular:
                    sorted_norms = s["model_norms"].sort()[0]
                    if irregular:
                        sorted_norms = sorted_norms[-k:]
                    num = sorted_norms.numel()
                    median = sorted_norms[min(num - 1, (num // 4) * 2)]
                    if not bool(torch.isfinite(median)):
                        raise RuntimeError("Too many grads were not finite")
                    s["fstate"][0] = cs * median * (2.0 if irregular else 1.0)
                    s["istate"][0] = 1
                    s["istate"][1] = 0
                if int(s["istate"][0]):
                    ans = torch.nan_to_num(torch.clamp(s["fstate"][0] / (tot_norm + 1.0e-20),
                                                       max=1.0), nan=0.0)
                    s["istate"][1] += int(ans < 1.0)
                    g.mul_(ans)
                    torch.nan_to_num_(g, nan=0.0, posinf=0.0, neginf=0.0)
            delta.mul_(beta1)
            s["scale_grads"][k % P] = self._seg_sum(s, p * g)
            if k % P == P - 1:
                s["param_rms"] = (self._seg_sum(s, p * p) / lens).sqrt()
                if k > 0:
                    rms = s["param_rms"]
                    sg = s["scale_grads"]
                    s["scale_exp_avg_sq"].mul_(h["beta2c"]).add_((sg * sg).mean(dim=0),
                                                                 alpha=1 - h["beta2c"])
                    denom = s["scale_exp_avg_sq"].sqrt() + eps
                    scale_step = -(lr * slr) * (h["bc2_size"] ** 0.5) * sg.sum(dim=0) / denom
                    scale_step = scale_step.masked_fill(rms < group["param_min_rms"], 0.0)
                    scale_step = torch.minimum(scale_step, (group["param_max_rms"] - rms) / rms)
                    scale_step = scale_step.masked_fill(scalar, 0.0)
                    delta.add_(p * scale_step[seg], alpha=(1 - beta1))
            eas.mul_(beta2).add_(g * g, alpha=1 - beta2)
            bc2 = h["bc2"]
            bc_vec = torch.where(scalar, torch.full_like(lens, bc2),
                                 torch.full_like(lens, bc2 if bc2 < 0.99 else 1.0))
            coef = torch.where(scalar, torch.full_like(lens, -lr * slr * (1 - beta1)),
                               -lr * (1 - beta1) * s["param_rms"].clamp(min=group["param_min_rms"]))
            denom = (eas / bc_vec[seg]).sqrt_().add_(eps)
            delta.add_(g / denom * coef[seg])
            if bool(scalar.any()):
                lim = torch.where(scalar, torch.full_like(lens, group["scalar_max"]),
                                  torch.full_like(lens, float("inf")))[seg]
                torch.minimum(p, lim, out=p)
                torch.maximum(p, -lim, out=p)
            p.add_(delta)
            s["step"] = k + 1
        if self.zero_grad_in_step:
            st.zero_grad()

    # ------------------------------------------------------------------ checkpoint interchange
    def _shape_batches(self, group, s):
        """The reference keeps its state per BATCH of same-shaped tensors, in the state of the
        batch's first parameter, batches ordered by (dtype, *shape) (optimizer/scaled_adam.py:
        63-101): -> [[(param, store index)]] in that order."""
        trainable = [p for p in group["params"] if p.requires_grad]
        by_key = {}
        for j, p in enumerate(trainable):
            by_key.setdefault((str(p.dtype), *p.shape), []).append((p, s["lo"] + j))
        return [by_key[k] for k in sorted(by_key)]

    def state_dict(self):
        """torch's optimizer state_dict layout with the reference's per-batch entries (`step`,
        `delta`, `exp_avg_sq`, `param_rms`, `scale_exp_avg_sq`, `scale_grads`; `model_norms`,
        `model_norm_threshold`, `num_clipped` on the group's first batch), gathered from the
        flat buffers: a reference checkpoint's optimizer state loads here and vice versa."""
        if self._gstate is None:
            self._init()
        st = self.store
        state, groups, start = {}, [], 0

        def rows(buf, batch):
            return torch.stack([buf[st.offsets[i]:st.offsets[i] + st.lengths[i]].view(p.shape)
                                for p, i in batch]).clone()

        for group, s in zip(self.param_groups, self._gstate):
            ps = group["params"]
            index = {id(p): start + j for j, p in enumerate(ps)}
            packed = {k: v for k, v in group.items() if k != "params"}
            packed["params"] = list(range(start, start + len(ps)))
            groups.append(packed)
            P = group["size_update_period"]
            for bi, batch in enumerate(self._shape_batches(group, s)):
                p0 = batch[0][0]
                e = {"step": s["step"], "delta": rows(self._delta, batch),
                     "exp_avg_sq": rows(self._eas, batch)}
                if p0.numel() > 1:
                    loc = torch.tensor([i - s["lo"] for _, i in batch], device=st.flat_p.device)
                    shp = (len(batch),) + (1,) * p0.dim()
                    e["param_rms"] = s["param_rms"][loc].view(shp).clone()
                    e["scale_exp_avg_sq"] = s["scale_exp_avg_sq"][loc].view(shp).clone()
                    e["scale_grads"] = s["scale_grads"][:, loc].reshape((P,) + shp).clone()
                if bi == 0 and group["clipping_scale"] is not None and s["step"] > 1:
                    e["model_norms"] = s["model_norms"].clone()
                    if int(s["istate"][0]):
                        e["model_norm_threshold"] = float(s["fstate"][0])
                        e["num_clipped"] = int(s["istate"][1])
                state[index[id(p0)]] = e
            start += len(ps)
        return {"state": state, "param_groups": groups}

    @torch.no_grad()
    def load_state_dict(self, sd):
        if self._gstate is None:
            self._init()
        st = self.store
        if len(sd["param_groups"]) != len(self.param_groups):
            raise ValueError("loaded state dict has a different number of parameter groups")
        start = 0
        for group, saved, s in zip(self.param_groups, sd["param_groups"], self._gstate):
            if len(saved["params"]) != len(group["params"]):
                raise ValueError("loaded state dict contains a parameter group that doesn't match "
                                 "the size of optimizer's group")
            for k, v in saved.items():
                if k != "params":
                    group[k] = v
            ps = group["params"]
            index = {id(p): start + j for j, p in enumerate(ps)}
            for bi, batch in enumerate(self._shape_batches(group, s)):
                e = sd["state"].get(index[id(batch[0][0])])
                if e is None:
                    e = sd["state"].get(str(index[id(batch[0][0])]))
                if e is None:
                    continue
                s["step"] = int(e["step"])
                for b, (p, i) in enumerate(batch):
                    o, n = st.offsets[i], st.lengths[i]
                    self._delta[o:o + n].copy_(e["delta"][b].reshape(-1))
                    self._eas[o:o + n].copy_(e["exp_avg_sq"][b].reshape(-1))
                    if "param_rms" in e:
                        j = i - s["lo"]
                        s["param_rms"][j] = e["param_rms"][b].reshape(())
                        s["scale_exp_avg_sq"][j] = e["scale_exp_avg_sq"][b].reshape(())
                        s["scale_grads"][:, j] = e["scale_grads"][:, b].reshape(-1)
                if bi == 0 and "model_norms" in e:
                    s["model_norms"].copy_(e["model_norms"])
                    if "model_norm_threshold" in e:
                        s["fstate"][0] = float(e["model_norm_threshold"])
                        s["istate"][0] = 1
                        s["istate"][1] = int(e.get("num_clipped", 0))
            start += len(ps)

    def zero_grad(self, set_to_none: bool = False):
        if self.store is not None:
            self.store.zero_grad()
        else:
            super().zero_grad(set_to_none=False)
