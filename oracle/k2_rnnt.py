"""Oracle: k2 pruned RNN-T pieces + torchaudio RNN-T loss, torch-CPU fp32 restatement.

PARITY UNPINNED: k2==1.24.3 (requirements.txt:4) and torchaudio==0.13.1
(requirements.txt:16) are absent from the reference tree and from this image, and the
reference's own tests only log loss values (model/loss/pruned_rnnt_loss_test.py:32-46,
model/loss/rnnt_loss_test.py:19-29).  What is restated here is the published k2 v1.24.3
algorithm (k2/python/k2/rnnt_loss.py: get_rnnt_logprobs_smoothed, rnnt_loss_smoothed,
get_rnnt_prune_ranges, _adjust_pruning_lower_bound, do_rnnt_pruning,
get_rnnt_logprobs_pruned, rnnt_loss_pruned; k2/python/k2/mutual_information.py) as
called from model/joiner/joiner.py:100-123 and model/loss/pruned_rnnt_loss.py:39-48,
anchored by self-checks in tests/ (brute-force path enumeration, pruned == unpruned
when s_range = S+1, simple == full on additive logits, range invariants, gradcheck).
"""
import ctypes
import os

import numpy as np
import torch

NEG_INF = float("-inf")


# ---------------------------------------------------------------- mutual information
def mutual_information_np(px, py, boundary):
    """px (B,S,T+1), py (B,S+1,T) float32 numpy; boundary (B,4) = (0,0,S_b,T_b).

    p[s,t] = logaddexp(p[s-1,t] + px[s-1,t], p[s,t-1] + py[s,t-1]), p[0,0] = 0
    returns p (B,S+1,T+1), ans (B,), px_grad, py_grad (grad of ans w.r.t. px/py).
    """
    px = np.asarray(px)
    py = np.asarray(py)
    dt = px.dtype
    B, S, T1 = px.shape
    T = py.shape[2]
    assert T1 == T + 1 and py.shape[1] == S + 1
    p = np.full((B, S + 1, T + 1), -np.inf, dtype=dt)
    ans = np.zeros(B, dtype=dt)
    gx = np.zeros_like(px)
    gy = np.zeros_like(py)
    for b in range(B):
        Sb, Tb = int(boundary[b, 2]), int(boundary[b, 3])
        pb = p[b]
        pb[0, 0] = 0.0
        with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
            for t in range(1, Tb + 1):
                pb[0, t] = pb[0, t - 1] + py[b, 0, t - 1]
            for s in range(1, Sb + 1):
                pb[s, 0] = pb[s - 1, 0] + px[b, s - 1, 0]
                up = pb[s - 1, 1:Tb + 1] + px[b, s - 1, 1:Tb + 1]
                pyrow = py[b, s, :Tb]
                cur = pb[s, 0]
                row = pb[s]
                for t in range(1, Tb + 1):
                    cur = np.logaddexp(up[t - 1], cur + pyrow[t - 1])
                    row[t] = cur
            ans[b] = pb[Sb, Tb]
            # backward: occupation counts
            pg = np.zeros((Sb + 2, Tb + 2), dtype=np.float64)
            pg[Sb, Tb] = 1.0
            for s in range(Sb, -1, -1):
                for t in range(Tb, -1, -1):
                    if s == Sb and t == Tb:
                        continue
                    xg = 0.0
                    yg = 0.0
                    if s < Sb and pg[s + 1, t] != 0.0 and np.isfinite(pb[s + 1, t]):
                        e = np.exp(np.float64(pb[s, t]) + px[b, s, t] - pb[s + 1, t])
                        xg = pg[s + 1, t] * e if np.isfinite(e) else 0.0
                    if t < Tb and pg[s, t + 1] != 0.0 and np.isfinite(pb[s, t + 1]):
                        e = np.exp(np.float64(pb[s, t]) + py[b, s, t] - pb[s, t + 1])
                        yg = pg[s, t + 1] * e if np.isfinite(e) else 0.0
                    if s < Sb:
                        gx[b, s, t] = xg
                    if t < Tb:
                        gy[b, s, t] = yg
                    pg[s, t] = xg + yg
    return p, ans, gx, gy


_CLIB = [None, False]


def _clib():
    """oracle/liboracle_c.so (oracle/csrc/mutual_info.c, built by oracle/build.py) or None."""
    if not _CLIB[1]:
        _CLIB[1] = True
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "liboracle_c.so")
        if os.path.exists(path) and os.environ.get("ORACLE_MI", "c") != "numpy":
            lib = ctypes.CDLL(path)
            lib.oracle_mutual_information.restype = ctypes.c_int
            lib.oracle_mutual_information.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + \
                [ctypes.c_void_p] * 4
            _CLIB[0] = lib
    return _CLIB[0]


def mutual_information_c(px, py, boundary):
    """Same contract as mutual_information_np, computed by the plain-C restatement."""
    lib = _clib()
    assert lib is not None, "oracle/liboracle_c.so is not built (python -m oracle.build)"
    px = np.ascontiguousarray(px, dtype=np.float32)
    py = np.ascontiguousarray(py, dtype=np.float32)
    bd = np.ascontiguousarray(boundary, dtype=np.int64)
    B, S, T1 = px.shape
    T = py.shape[2]
    assert T1 == T + 1 and py.shape[1] == S + 1
    p = np.empty((B, S + 1, T + 1), np.float32)
    ans = np.empty(B, np.float32)
    gx = np.empty_like(px)
    gy = np.empty_like(py)
    rc = lib.oracle_mutual_information(px.ctypes.data, py.ctypes.data, bd.ctypes.data, B, S, T,
                                       p.ctypes.data, ans.ctypes.data, gx.ctypes.data,
                                       gy.ctypes.data)
    assert rc == 0
    return p, ans, gx, gy


def mutual_information_any(px, py, boundary):
    """C restatement when built (fast enough for the C3-size CPU baseline), else numpy."""
    if _clib() is not None and np.asarray(px).dtype == np.float32:
        return mutual_information_c(px, py, boundary)
    return mutual_information_np(px, py, boundary)


class _MutualInformation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, px, py, boundary):
        _, ans, gx, gy = mutual_information_any(px.detach().numpy(), py.detach().numpy(),
                                                boundary.numpy())
        ctx.save_for_backward(torch.from_numpy(gx), torch.from_numpy(gy))
        return torch.from_numpy(ans)

    @staticmethod
    def backward(ctx, g):
        gx, gy = ctx.saved_tensors
        return gx * g.reshape(-1, 1, 1), gy * g.reshape(-1, 1, 1), None


def mutual_information_recursion(px, py, boundary, return_grad=False):
    if return_grad:
        _, ans, gx, gy = mutual_information_any(px.detach().numpy(), py.detach().numpy(),
                                                boundary.numpy())
        scores = _MutualInformation.apply(px, py, boundary)
        return scores, (torch.from_numpy(gx), torch.from_numpy(gy))
    return _MutualInformation.apply(px, py, boundary)


# ---------------------------------------------------------------- simple (smoothed) loss
def fix_for_boundary(px, boundary):
    B, S, T1 = px.shape
    idx = boundary[:, 3].reshape(B, 1, 1).expand(B, S, 1)
    return px.scatter(2, idx, NEG_INF)


def get_rnnt_logprobs_smoothed(lm, am, symbols, termination_symbol, boundary,
                               lm_only_scale=0.0, am_only_scale=0.0):
    """k2.get_rnnt_logprobs_smoothed, rnnt_type="regular".  With both scales 0 (the only
    setting the reference uses, model/joiner/joiner.py:22-23,105-106) k2 replaces them by
    1e-20, so the lm-only / am-only terms vanish in fp32; they are kept for other scales."""
    B, T, C = am.shape
    S = lm.shape[1] - 1
    am_max = am.max(dim=2, keepdim=True)[0]
    lm_max = lm.max(dim=2, keepdim=True)[0]
    am_probs = (am - am_max).exp()
    lm_probs = (lm - lm_max).exp()
    tiny = torch.finfo(lm_probs.dtype).tiny
    normalizers = (torch.matmul(lm_probs, am_probs.transpose(1, 2)) + tiny).log()
    lmonly_normalizers = lm_probs.sum(dim=2, keepdim=True)
    unigram_lm = torch.mean(lm_probs / lmonly_normalizers, dim=(0, 1), keepdim=True) + tiny
    amonly_normalizers = (torch.mv(am_probs.reshape(-1, C), unigram_lm.reshape(C))
                          .reshape(B, T, 1).log() + am_max).transpose(1, 2)
    unigram_lm = unigram_lm.log()
    lmonly_normalizers = lmonly_normalizers.log() + lm_max
    normalizers = normalizers + lm_max + am_max.transpose(1, 2)
    px_am = torch.gather(am.unsqueeze(1).expand(B, S, T, C), 3,
                         symbols.reshape(B, S, 1, 1).expand(B, S, T, 1)).squeeze(-1)
    px_am = torch.cat((px_am, torch.full((B, S, 1), NEG_INF, dtype=px_am.dtype)), dim=2)
    px_lm = torch.gather(lm[:, :S], 2, symbols.unsqueeze(-1))
    px_lm_unigram = torch.gather(unigram_lm.expand(B, S, C), 2, symbols.unsqueeze(-1))
    px = px_am + px_lm
    px = torch.cat((px[:, :, :T] - normalizers[:, :S, :], px[:, :, T:]), dim=2)
    px_amonly = px_am + px_lm_unigram
    px_amonly = torch.cat((px_amonly[:, :, :T] - amonly_normalizers, px_amonly[:, :, T:]), dim=2)
    px_lmonly = px_lm - lmonly_normalizers[:, :S, :]
    py_am = am[:, :, termination_symbol].unsqueeze(1)
    py_lm = lm[:, :, termination_symbol].unsqueeze(2)
    py = py_am + py_lm - normalizers
    py_amonly = py_am + unigram_lm[0][0][termination_symbol] - amonly_normalizers
    py_lmonly = py_lm - lmonly_normalizers
    combined = 1.0 - lm_only_scale - am_only_scale
    if lm_only_scale == 0.0:
        lm_only_scale = 1.0e-20
    if am_only_scale == 0.0:
        am_only_scale = 1.0e-20
    px_i = px * combined + px_lmonly * lm_only_scale + px_amonly * am_only_scale
    py_i = py * combined + py_lmonly * lm_only_scale + py_amonly * am_only_scale
    return fix_for_boundary(px_i, boundary), py_i


def rnnt_loss_smoothed(lm, am, symbols, termination_symbol, boundary, lm_only_scale=0.0,
                       am_only_scale=0.0, reduction="mean", return_grad=True):
    px, py = get_rnnt_logprobs_smoothed(lm, am, symbols, termination_symbol, boundary,
                                        lm_only_scale, am_only_scale)
    out = mutual_information_recursion(px, py, boundary, return_grad=return_grad)
    scores = out[0] if return_grad else out
    loss = {"mean": lambda s: -s.mean(), "sum": lambda s: -s.sum(), "none": lambda s: -s}[
        reduction](scores)
    return (loss, out[1]) if return_grad else loss


# ---------------------------------------------------------------- pruning
def monotonic_lower_bound(x):
    """ans[..., i] = min_{j >= i} x[..., j] (k2.monotonic_lower_bound)."""
    return torch.flip(torch.cummin(torch.flip(x, [-1]), dim=-1)[0], [-1])


def _adjust_pruning_lower_bound(s_begin, s_range):
    B, T = s_begin.shape
    ar = (s_range - 1) * torch.arange(T)
    s_begin = monotonic_lower_bound(s_begin)
    s_begin = -(s_begin - ar)
    s_begin = monotonic_lower_bound(s_begin)
    s_begin = torch.clamp(s_begin, min=0)
    return -(s_begin - ar)


def get_rnnt_prune_ranges(px_grad, py_grad, boundary, s_range):
    B, S, T1 = px_grad.shape
    T = py_grad.shape[-1]
    S1 = S + 1
    assert T1 == T + 1 and S >= 1
    if s_range > S:
        s_range = S + 1
    assert s_range >= 2
    blk = torch.stack([py_grad[:, i:i + S1 - s_range + 1, :] for i in range(s_range)], dim=2)
    blk_sum = blk.sum(dim=2)
    px_pad = torch.cat((torch.zeros(B, 1, T1, dtype=px_grad.dtype), px_grad), dim=1)
    final = blk_sum - px_pad[:, :S1 - s_range + 1, :T]
    s_begin = torch.argmax(final, dim=1)
    mask = torch.arange(T).reshape(1, T).expand(B, T) < boundary[:, 3].reshape(B, 1) - 1
    pad = torch.clamp(boundary[:, 2].reshape(B, 1) - s_range + 1, min=0)
    s_begin = torch.where(mask, s_begin, pad)
    s_begin = _adjust_pruning_lower_bound(s_begin, s_range)
    return s_begin.reshape(B, T, 1).expand(B, T, s_range) + torch.arange(s_range)


def do_rnnt_pruning(am, lm, ranges):
    B, T, R = ranges.shape
    C = am.shape[-1]
    S1 = lm.shape[1]
    am_p = am.unsqueeze(2).expand(B, T, R, C)
    lm_p = torch.gather(lm.unsqueeze(1).expand(B, T, S1, C), 2,
                        ranges.reshape(B, T, R, 1).expand(B, T, R, C))
    return am_p, lm_p


def _roll_by_shifts(src, shifts):
    B, T, S = src.shape
    idx = (torch.arange(S).view(1, 1, S).repeat(B, T, 1) - shifts.reshape(B, T, 1)) % S
    return torch.gather(src, 2, idx)


def get_rnnt_logprobs_pruned(logits, symbols, ranges, termination_symbol, boundary):
    B, T, R, C = logits.shape
    S = symbols.shape[1]
    normalizers = torch.logsumexp(logits, dim=3)
    sym_t = torch.cat((symbols, torch.full((B, 1), termination_symbol, dtype=symbols.dtype)), 1)
    pruned_symbols = torch.gather(sym_t.unsqueeze(1).expand(B, T, S + 1), 2, ranges)
    px = torch.gather(logits, 3, pruned_symbols.reshape(B, T, R, 1)).squeeze(-1) - normalizers
    px = torch.cat((px, torch.full((B, T, S + 1 - R), NEG_INF, dtype=px.dtype)), dim=2)
    px = _roll_by_shifts(px, ranges[:, :, 0])[:, :, :S].permute(0, 2, 1)
    px = torch.cat((px, torch.full((B, S, 1), NEG_INF, dtype=px.dtype)), dim=2)
    py = logits[:, :, :, termination_symbol] - normalizers
    py = torch.cat((py, torch.full((B, T, S + 1 - R), NEG_INF, dtype=py.dtype)), dim=2)
    py = _roll_by_shifts(py, ranges[:, :, 0]).permute(0, 2, 1)
    return fix_for_boundary(px, boundary), py


def rnnt_loss_pruned(logits, symbols, ranges, termination_symbol, boundary, reduction="mean"):
    px, py = get_rnnt_logprobs_pruned(logits, symbols, ranges, termination_symbol, boundary)
    scores = mutual_information_recursion(px.contiguous(), py.contiguous(), boundary)
    return {"mean": lambda s: -s.mean(), "sum": lambda s: -s.sum(), "none": lambda s: -s}[
        reduction](scores)


# ---------------------------------------------------------------- joiner prune path
def joiner_pruned(am, lm, target, target_lengths, encoder_out_lengths, prune_range,
                  activation="relu", blank=0):
    """model/joiner/joiner.py:74-124,148-178 with use_out_project=False.
    am = _enc_proj(encoder_out) (B,T,C); lm = _pre_proj(predict_out) (B,S+1,C)."""
    B = am.shape[0]
    boundary = torch.zeros((B, 4), dtype=torch.int64)
    boundary[:, 2] = target_lengths
    boundary[:, 3] = encoder_out_lengths
    simple_loss, (px_grad, py_grad) = rnnt_loss_smoothed(lm.float(), am.float(), target, blank,
                                                          boundary)
    ranges = get_rnnt_prune_ranges(px_grad, py_grad, boundary, prune_range)
    am_p, lm_p = do_rnnt_pruning(am, lm, ranges)
    act = torch.relu if activation == "relu" else torch.tanh
    return act(am_p + lm_p), boundary, ranges, simple_loss


# ---------------------------------------------------------------- torchaudio RNNTLoss
def rnnt_loss_full(logits, targets, logit_lengths, target_lengths, blank=0, reduction="mean"):
    """torchaudio.transforms.RNNTLoss(blank, clamp=-1, reduction) as called at
    model/loss/rnnt_loss.py:42-44: log_softmax inside, -log P(y|x), mean over batch.
    Same lattice recursion as above (p[U_b][T_b] includes the final blank)."""
    B, T, U1, V = logits.shape
    S = U1 - 1
    lp = torch.log_softmax(logits.float(), dim=-1)
    boundary = torch.zeros((B, 4), dtype=torch.int64)
    boundary[:, 2] = target_lengths.long()
    boundary[:, 3] = logit_lengths.long()
    tg = targets.long()
    px = torch.gather(lp[:, :, :S, :], 3, tg.reshape(B, 1, S, 1).expand(B, T, S, 1)).squeeze(-1)
    px = px.permute(0, 2, 1)
    px = torch.cat((px, torch.full((B, S, 1), NEG_INF)), dim=2)
    px = fix_for_boundary(px, boundary)
    py = lp[:, :, :, blank].permute(0, 2, 1)
    scores = mutual_information_recursion(px.contiguous(), py.contiguous(), boundary)
    return {"mean": lambda s: -s.mean(), "sum": lambda s: -s.sum(), "none": lambda s: -s}[
        reduction](scores)
