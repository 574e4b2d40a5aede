"""torch.autograd wrappers around the C ABI of libs2t_mi355.so.

Everything here runs on the HIP stream torch considers current; torch is used for
device memory and autograd plumbing only.  There is no CPU path: CPU tensors raise.
"""
import math

import numpy as np
import torch

from . import _native as N

_NEG_INF = float("-inf")
CTC_MAX_LABELS = 1023     # == S2T_CTC_MAX_LABELS (include/s2t_mi355.h)


def _dev_check(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("speech2text_amd kernels run on the GPU only (no CPU fallback)")


# =============================================================== fbank
class FbankTables:
    """Host-built constant tables of the kaldi fbank (window, FFT twiddles, mel banks).

    Formulas follow dataset/frontend/frontend.py:85-94 -> kaldi.fbank as stated by
    sample_data/model/frontend.script (povey window, mel scale 1127 ln(1+f/700),
    low 20 Hz, high = nyquist + high_freq if high_freq <= 0)."""

    def __init__(self, num_mel_bins=80, sample_frequency=16000.0, low_freq=20.0, high_freq=0.0,
                 device="cuda"):
        nfft = 512
        win = torch.hann_window(400, periodic=False, dtype=torch.float32).pow(0.85)
        m = np.arange(nfft, dtype=np.float64)
        tw = np.stack([np.cos(2 * math.pi * m / nfft), -np.sin(2 * math.pi * m / nfft)], 1)
        nyq = 0.5 * sample_frequency
        hf = high_freq + nyq if high_freq <= 0.0 else high_freq
        mel_lo = 1127.0 * math.log(1.0 + low_freq / 700.0)
        mel_hi = 1127.0 * math.log(1.0 + hf / 700.0)
        delta = (mel_hi - mel_lo) / (num_mel_bins + 1)
        b = torch.arange(num_mel_bins, dtype=torch.float32).unsqueeze(1)
        left = b * delta + mel_lo
        center = (b + 1.0) * delta + mel_lo
        right = (b + 2.0) * delta + mel_lo
        freq = torch.arange(nfft // 2, dtype=torch.float32) * (sample_frequency / nfft)
        mel = ((freq / 700.0 + 1.0).log() * 1127.0).unsqueeze(0)
        up = (mel - left) / (center - left)
        down = (right - mel) / (right - center)
        w = torch.clamp(torch.min(up, down), min=0.0).numpy()          # (M, 256)
        off, k0, vals = [0], [], []
        for r in range(num_mel_bins):
            nz = np.nonzero(w[r])[0]
            if nz.size == 0:
                k0.append(0)
            else:
                k0.append(int(nz[0]))
                vals.extend(w[r, nz[0]:nz[-1] + 1].tolist())
            off.append(len(vals))
        self.num_mel = num_mel_bins
        self.nnz = len(vals)
        self.dense = w
        self.window = win.to(device)
        self.twiddle = torch.from_numpy(tw.astype(np.float32)).contiguous().to(device)
        self.mel_off = torch.tensor(off, dtype=torch.int32, device=device)
        self.mel_k0 = torch.tensor(k0, dtype=torch.int32, device=device)
        self.mel_w = torch.tensor(vals if vals else [0.0], dtype=torch.float32, device=device)


def fbank_batch(pcm, num_samples, tables, cmvn_mean=None, cmvn_istd=None, scale_in=1.0,
                max_frames=None):
    """pcm (B,Nmax) f32 device, num_samples (B,) i64 device -> feats (B,n_max,M), n_frames (B,)."""
    _dev_check(pcm, num_samples)
    B, nmax = pcm.shape
    if max_frames is None:
        max_frames = 0 if nmax < 400 else 1 + (nmax - 400) // 160
    out = torch.empty((B, max_frames, tables.num_mel), dtype=torch.float32, device=pcm.device)
    frames = torch.empty((B,), dtype=torch.int64, device=pcm.device)
    if B == 0 or max_frames == 0:
        return out, frames.zero_()
    N.PROF[0] and N.profile_note("s2t_fbank_f32", 4.0 * (pcm.numel() + out.numel()))
    rc = N.lib().s2t_fbank_f32(N.fp(pcm), pcm.stride(0), N.lp(num_samples), B,
                               N.fp(tables.window), N.fp(tables.twiddle), N.ip(tables.mel_off),
                               N.ip(tables.mel_k0), N.fp(tables.mel_w), tables.nnz,
                               tables.num_mel, 1.1920928955078125e-07, float(scale_in),
                               N.fp(cmvn_mean), N.fp(cmvn_istd), N.fp(out), max_frames,
                               N.lp(frames), N.stream())
    N.check(rc, "s2t_fbank_f32")
    return out, frames


# =============================================================== CTC
class _CtcLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, in_len, tgt_len, blank, reduction, zero_infinity):
        _dev_check(logits)
        logits = logits.contiguous().float()
        B, T, V = logits.shape
        dev = logits.device
        targets = targets.to(device=dev, dtype=torch.int64).contiguous()
        in_len = in_len.to(device=dev, dtype=torch.int64).contiguous()
        tgt_len = tgt_len.to(device=dev, dtype=torch.int64).contiguous()
        U = targets.shape[1] if targets.dim() == 2 else 0
        if U == 0:
            targets = torch.zeros((B, 1), dtype=torch.int64, device=dev)
        tl = tgt_len.clamp(min=1).float()
        if reduction == "mean":
            scale = 1.0 / (tl * B)
        else:
            scale = torch.ones(B, device=dev)
        scale = scale.contiguous()
        if U > CTC_MAX_LABELS:
            raise ValueError(f"CTC: padded label width {U} exceeds the kernel's limit of "
                             f"{CTC_MAX_LABELS} labels per utterance (2U+1 lattice states are "
                             "held in one workgroup's LDS, csrc/ctc.hip)")
        ws = torch.empty(N.lib().s2t_ctc_workspace_floats(B, T, U), dtype=torch.float32,
                         device=dev)
        per = torch.empty(B, dtype=torch.float32, device=dev)
        grad = torch.empty_like(logits)
        # 2 B T V 4 (logits in, gradient out) + 3 B T (2U+1) 4 (lattice rows)
        N.PROF[0] and N.profile_note("s2t_ctc_loss_fwd_bwd", 8.0 * B * T * V + 12.0 * B * T * (2 * U + 1))
        rc = N.lib().s2t_ctc_loss_fwd_bwd(N.fp(logits), N.lp(targets), targets.stride(0),
                                          N.lp(in_len), N.lp(tgt_len), B, T, V, U, int(blank),
                                          int(bool(zero_infinity)), N.fp(scale), N.fp(ws),
                                          N.fp(per), N.fp(grad), N.stream())
        N.check(rc, "s2t_ctc_loss_fwd_bwd")
        ctx.save_for_backward(grad)
        ctx.reduction = reduction
        if reduction == "mean":
            return (per / tl).mean()
        if reduction == "sum":
            return per.sum()
        return per

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        if ctx.reduction == "none":
            return grad * g.reshape(-1, 1, 1), None, None, None, None, None, None
        return grad * g, None, None, None, None, None, None


def ctc_loss(logits, targets, logits_length, targets_length, blank=0, reduction="mean",
             zero_infinity=True):
    """Fused log_softmax + CTC on batch-major logits (B,T,V)."""
    return _CtcLoss.apply(logits, targets, logits_length, targets_length, blank, reduction,
                          zero_infinity)


# =============================================================== RNN-T lattice
def _i64(t, dev):
    return t.to(device=dev, dtype=torch.int64).contiguous()


def mutual_information(px, py, boundary, want_grads=True):
    """Raw recursion: returns (scores (B,), p, px_grad, py_grad) -- no autograd."""
    B, S, T1 = px.shape
    T = T1 - 1
    dev = px.device
    p = torch.empty((B, S + 1, T + 1), dtype=torch.float32, device=dev)
    ans = torch.empty((B,), dtype=torch.float32, device=dev)
    L = N.lib()
    N.PROF[0] and N.profile_note("s2t_mutual_info_fwd", 4.0 * (px.numel() + py.numel() + p.numel()))
    N.check(L.s2t_mutual_info_fwd(N.fp(px), N.fp(py), N.lp(boundary), B, S, T, N.fp(p),
                                  N.fp(ans), N.stream()), "s2t_mutual_info_fwd")
    if not want_grads:
        return ans, p, None, None
    gx = torch.empty_like(px)
    gy = torch.empty_like(py)
    N.PROF[0] and N.profile_note("s2t_mutual_info_bwd", 4.0 * (2 * px.numel() + 2 * py.numel() + p.numel()))
    N.check(L.s2t_mutual_info_bwd(N.fp(px), N.fp(py), N.lp(boundary), N.fp(p), None, B, S, T,
                                  N.fp(gx), N.fp(gy), N.stream()), "s2t_mutual_info_bwd")
    return ans, p, gx, gy


class _SimpleRnntLoss(torch.autograd.Function):
    """k2.rnnt_loss_smoothed(lm, am, symbols, blank, lm_only_scale=0, am_only_scale=0,
    boundary, reduction='none', return_grad=True) -> (neg scores (B,), px_grad, py_grad)."""

    @staticmethod
    def forward(ctx, lm, am, symbols, boundary, blank):
        _dev_check(lm, am)
        lm = lm.contiguous().float()
        am = am.contiguous().float()
        B, T, C = am.shape
        S = lm.shape[1] - 1
        dev = am.device
        L = N.lib()
        st = N.stream()
        am_p = torch.empty_like(am)
        lm_p = torch.empty_like(lm)
        am_max = torch.empty((B, T), dtype=torch.float32, device=dev)
        lm_max = torch.empty((B, S + 1), dtype=torch.float32, device=dev)
        N.PROF[0] and N.profile_note("s2t_rnnt_row_exp", 8.0 * (am.numel() + lm.numel()) + 4.0 * B * (T + S + 1))
        N.check(L.s2t_rnnt_row_exp(N.fp(am), B * T, C, N.fp(am_p), N.fp(am_max), st), "row_exp")
        N.check(L.s2t_rnnt_row_exp(N.fp(lm), B * (S + 1), C, N.fp(lm_p), N.fp(lm_max), st),
                "row_exp")
        # the normaliser product exp(lm) exp(am)^T (reference model/joiner/joiner.py:100-108 ->
        # k2.rnnt_loss_smoothed) and its two gradients below: our batched kernel (csrc/gemm.hip), no
        # library launch on the loss path
        from . import zip_kernels as zk
        with zk.gemm_class(zk.CLS_F):
            nrm = zk.batched_matmul(0, lm_p, am_p)                       # (B,S+1,T)
        px = torch.empty((B, S, T + 1), dtype=torch.float32, device=dev)
        py = torch.empty((B, S + 1, T), dtype=torch.float32, device=dev)
        # (two gathered operands per lattice arc -- am / lm at the arc's symbol or at blank -- the normaliser once)
        N.PROF[0] and N.profile_note("s2t_rnnt_simple_pxpy", 4.0 * (3 * px.numel() + 3 * py.numel() + nrm.numel()))
        N.check(L.s2t_rnnt_simple_pxpy(N.fp(am), N.fp(lm), N.fp(am_max), N.fp(lm_max),
                                       N.fp(nrm), N.lp(symbols), N.lp(boundary), B, S, T, C,
                                       int(blank), N.fp(px), N.fp(py), st), "simple_pxpy")
        ans, _, gx, gy = mutual_information(px, py, boundary)
        ctx.save_for_backward(am_p, lm_p, nrm, gx, gy, symbols)
        ctx.blank = int(blank)
        ctx.mark_non_differentiable(gx, gy)
        return -ans, gx, gy

    @staticmethod
    def backward(ctx, g_loss, _gx, _gy):
        am_p, lm_p, nrm, gx, gy, symbols = ctx.saved_tensors
        B, T, C = am_p.shape
        S = lm_p.shape[1] - 1
        L = N.lib()
        st = N.stream()
        gscale = (-g_loss).contiguous().float()       # d loss / d score = -g
        W = torch.empty_like(nrm)
        N.PROF[0] and N.profile_note("s2t_rnnt_simple_w", 4.0 * (gx.numel() + gy.numel() + 2 * nrm.numel()))
        N.check(L.s2t_rnnt_simple_w(N.fp(gx), N.fp(gy), N.fp(nrm), N.fp(gscale), B, S, T,
                                    N.fp(W), st), "simple_w")
        from . import zip_kernels as zk
        with zk.gemm_class(zk.CLS_D):
            G_am = zk.batched_matmul(2, W, lm_p, own_tn=True)     # W^T @ exp(lm): (B,T,C)
            G_lm = zk.batched_matmul(1, W, am_p)                  # W @ exp(am):   (B,S+1,C)
        d_am = torch.empty_like(am_p)
        d_lm = torch.empty_like(lm_p)
        N.PROF[0] and N.profile_note("s2t_rnnt_simple_bwd", 12.0 * (am_p.numel() + lm_p.numel())
                                     + 4.0 * (gx.numel() + gy.numel()))
        N.check(L.s2t_rnnt_simple_bwd(N.fp(am_p), N.fp(lm_p), N.fp(G_am), N.fp(G_lm), N.fp(gx),
                                      N.fp(gy), N.fp(gscale), N.lp(symbols), B, S, T, C,
                                      ctx.blank, N.fp(d_am), N.fp(d_lm), 0, st), "simple_bwd")
        return d_lm, d_am, None, None, None


def rnnt_simple_loss(lm, am, symbols, boundary, blank=0):
    return _SimpleRnntLoss.apply(lm, am, symbols, boundary, blank)


def rnnt_prune_ranges(px_grad, py_grad, boundary, s_range):
    B, S, T1 = px_grad.shape
    T = T1 - 1
    if s_range > S:
        s_range = S + 1
    ranges = torch.empty((B, T, s_range), dtype=torch.int64, device=px_grad.device)
    N.PROF[0] and N.profile_note("s2t_rnnt_prune_ranges", 4.0 * (px_grad.numel() + py_grad.numel()) + 8.0 * ranges.numel())
    N.check(N.lib().s2t_rnnt_prune_ranges(N.fp(px_grad), N.fp(py_grad), N.lp(boundary), B, S, T,
                                          int(s_range), N.lp(ranges), N.stream()),
            "s2t_rnnt_prune_ranges")
    return ranges


_ACT = {"relu": 0, "tanh": 1}


class _PrunedJoinerLoss(torch.autograd.Function):
    """Fused Joiner tail + k2.rnnt_loss_pruned(reduction='none'): per-utterance neg score
    of the lattice logits[b,t,i,:] = act(am[b,t,:] + lm[b,ranges[b,t,0]+i,:])."""

    @staticmethod
    def forward(ctx, am, lm, ranges, symbols, boundary, blank, act):
        _dev_check(am, lm)
        am = am.contiguous().float()
        lm = lm.contiguous().float()
        B, T, C = am.shape
        S = lm.shape[1] - 1
        R = ranges.shape[2]
        dev = am.device
        L = N.lib()
        st = N.stream()
        px = torch.empty((B, S, T + 1), dtype=torch.float32, device=dev)
        py = torch.empty((B, S + 1, T), dtype=torch.float32, device=dev)
        lse = torch.empty((B, T, R), dtype=torch.float32, device=dev)
        N.PROF[0] and N.profile_note("s2t_rnnt_pruned_fwd",
                       4.0 * (am.numel() + lm.numel() + px.numel() + py.numel() + lse.numel())
                       + 8.0 * ranges.numel())
        N.check(L.s2t_rnnt_pruned_fwd(N.fp(am), N.fp(lm), N.lp(ranges), N.lp(symbols),
                                      N.lp(boundary), B, S, T, C, R, int(blank), _ACT[act],
                                      N.fp(px), N.fp(py), N.fp(lse), st), "pruned_fwd")
        ans, _, gx, gy = mutual_information(px, py, boundary)
        ctx.save_for_backward(am, lm, ranges, symbols, lse, gx, gy)
        ctx.blank, ctx.act = int(blank), _ACT[act]
        return -ans

    @staticmethod
    def backward(ctx, g):
        am, lm, ranges, symbols, lse, gx, gy = ctx.saved_tensors
        B, T, C = am.shape
        S = lm.shape[1] - 1
        R = ranges.shape[2]
        gscale = (-g).contiguous().float()
        d_am = torch.empty_like(am)
        d_lm = torch.empty_like(lm)
        # the (B,T,R,C) pruned logits are recomputed, never stored: the algorithmic traffic is the
        # operands and their gradients; the recomputation costs 2 B T R C flops of VALU per pass
        N.PROF[0] and N.profile_note("s2t_rnnt_pruned_bwd",
                       4.0 * (2 * am.numel() + 2 * lm.numel() + lse.numel() + gx.numel() + gy.numel())
                       + 8.0 * ranges.numel(), 4.0 * B * T * R * C)
        N.check(N.lib().s2t_rnnt_pruned_bwd(N.fp(am), N.fp(lm), N.lp(ranges), N.lp(symbols),
                                            N.fp(lse), N.fp(gx), N.fp(gy), N.fp(gscale), B, S, T,
                                            C, R, ctx.blank, ctx.act, N.fp(d_am), N.fp(d_lm), 0,
                                            N.stream()), "pruned_bwd")
        return d_am, d_lm, None, None, None, None, None


def rnnt_pruned_joiner_loss(am, lm, ranges, symbols, boundary, blank=0, activation="relu"):
    return _PrunedJoinerLoss.apply(am, lm, ranges, symbols, boundary, blank, activation)


class _LatticeLoss(torch.autograd.Function):
    """Materialised lattice: logits (B,T,R,V); ranges None => full lattice (R = S+1)."""

    @staticmethod
    def forward(ctx, logits, ranges, symbols, boundary, blank):
        _dev_check(logits)
        logits = logits.contiguous().float()
        B, T, R, V = logits.shape
        S = symbols.shape[1]
        dev = logits.device
        px = torch.empty((B, S, T + 1), dtype=torch.float32, device=dev)
        py = torch.empty((B, S + 1, T), dtype=torch.float32, device=dev)
        lse = torch.empty((B, T, R), dtype=torch.float32, device=dev)
        N.check(N.lib().s2t_rnnt_lattice_fwd(N.fp(logits), N.lp(ranges), N.lp(symbols),
                                             N.lp(boundary), B, S, T, V, R, int(blank), N.fp(px),
                                             N.fp(py), N.fp(lse), N.stream()), "lattice_fwd")
        ans, _, gx, gy = mutual_information(px, py, boundary)
        ctx.save_for_backward(logits, symbols, lse, gx, gy)
        ctx.ranges = ranges
        ctx.blank = int(blank)
        return -ans

    @staticmethod
    def backward(ctx, g):
        logits, symbols, lse, gx, gy = ctx.saved_tensors
        B, T, R, V = logits.shape
        S = symbols.shape[1]
        gscale = (-g).contiguous().float()
        d = torch.empty_like(logits)
        N.check(N.lib().s2t_rnnt_lattice_bwd(N.fp(logits), N.lp(ctx.ranges), N.lp(symbols),
                                             N.fp(lse), N.fp(gx), N.fp(gy), N.fp(gscale), B, S, T,
                                             V, R, ctx.blank, N.fp(d), N.stream()),
                "lattice_bwd")
        return d, None, None, None, None


def rnnt_lattice_loss(logits, ranges, symbols, boundary, blank=0):
    return _LatticeLoss.apply(logits, ranges, symbols, boundary, blank)


def make_boundary(target_lengths, frame_lengths, device):
    B = target_lengths.shape[0]
    b = torch.zeros((B, 4), dtype=torch.int64, device=device)
    b[:, 2] = target_lengths.to(device)
    b[:, 3] = frame_lengths.to(device)
    return b


# =============================================================== BEST-RQ SSL heads
class _SmoothedNll(torch.autograd.Function):
    """Per-row  C0 - sum_c t_c log_softmax(scale * logits)_c  (csrc/ssl_loss.hip)."""

    @staticmethod
    def forward(ctx, logits, labels, scale, t_other, t_label, c0):
        _dev_check(logits)
        logits = logits.contiguous().float()
        rows, K = logits.shape
        labels = labels.to(device=logits.device, dtype=torch.int64).contiguous()
        row = torch.empty(rows, dtype=torch.float32, device=logits.device)
        lse = torch.empty(rows, dtype=torch.float32, device=logits.device)
        N.PROF[0] and N.profile_note("s2t_smoothed_nll_fwd", 4.0 * logits.numel())
        N.check(N.lib().s2t_smoothed_nll_fwd(N.fp(logits), N.lp(labels), rows, K, float(scale),
                                             float(t_other), float(t_label), float(c0),
                                             N.fp(row), N.fp(lse), N.stream()), "smoothed_nll_fwd")
        ctx.save_for_backward(logits, labels, lse)
        ctx.cfg = (float(scale), float(t_other), float(t_label))
        return row

    @staticmethod
    def backward(ctx, g):
        logits, labels, lse = ctx.saved_tensors
        scale, t_other, t_label = ctx.cfg
        rows, K = logits.shape
        grad = torch.empty_like(logits)
        N.PROF[0] and N.profile_note("s2t_smoothed_nll_bwd", 8.0 * logits.numel())
        N.check(N.lib().s2t_smoothed_nll_bwd(N.fp(logits), N.lp(labels), N.fp(lse),
                                             N.fp(g.contiguous().float()), rows, K, scale, t_other,
                                             t_label, N.fp(grad), N.stream()), "smoothed_nll_bwd")
        return grad, None, None, None, None, None


def smoothed_nll_rows(logits2d, labels, scale, t_other, t_label, c0):
    return _SmoothedNll.apply(logits2d, labels, scale, t_other, t_label, c0)


# =============================================================== StatelessPredictor context
PRED_MAX_CONTEXT = 8      # == PRED_MAXK (csrc/predictor.hip)


class _PredictorContext(torch.autograd.Function):
    """out[b,u,:] = sum_k w[:,k] * emb[tokens[b,u+k]]: nn.Embedding + depthwise nn.Conv1d of
    reference stateless_predictor.py:27-105 as one gather pass (csrc/predictor.hip)."""

    @staticmethod
    def forward(ctx, tokens, emb, w):
        _dev_check(tokens, emb, w)
        B, L = tokens.shape
        V, D = emb.shape
        K = w.shape[-1]
        tok = tokens.to(torch.int32).contiguous()
        e, wk = emb.contiguous().float(), w.reshape(D, K).contiguous().float()
        out = torch.empty((B, L - K + 1, D), dtype=torch.float32, device=emb.device)
        N.PROF[0] and N.profile_note("s2t_predictor_ctx_fwd", 4.0 * (K + 1) * out.numel() + 4.0 * tok.numel())
        N.check(N.lib().s2t_predictor_ctx_fwd(N.ip(tok), N.fp(e), N.fp(wk), B, L, K, D, V, N.fp(out),
                                              N.stream()), "s2t_predictor_ctx_fwd")
        ctx.save_for_backward(tok, e, wk)
        ctx.wshape = w.shape
        return out

    @staticmethod
    def backward(ctx, g):
        tok, e, wk = ctx.saved_tensors
        B, L = tok.shape
        V, D = e.shape
        K = wk.shape[1]
        g = g.contiguous().float()
        de = torch.zeros_like(e) if ctx.needs_input_grad[1] else None
        dw = torch.zeros_like(wk) if ctx.needs_input_grad[2] else None
        N.PROF[0] and N.profile_note("s2t_predictor_ctx_bwd", 4.0 * (2 * K + 1) * g.numel() + 4.0 * tok.numel())
        N.check(N.lib().s2t_predictor_ctx_bwd(N.ip(tok), N.fp(e), N.fp(wk), N.fp(g), B, L, K, D, V,
                                              N.fp(de), N.fp(dw), N.stream()), "s2t_predictor_ctx_bwd")
        return None, de, (None if dw is None else dw.view(ctx.wshape))


def predictor_context(tokens, emb_weight, conv_weight):
    """tokens (B, L) integer, emb_weight (V, D), conv_weight (D, 1, K) -> (B, L-K+1, D)."""
    return _PredictorContext.apply(tokens, emb_weight, conv_weight)

