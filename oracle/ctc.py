"""Oracle: CTC loss + gradient w.r.t. the (un-normalised) logits, numpy float64.

Restates model/loss/ctc_loss.py:35-41: F.log_softmax(logits) -> transpose ->
nn.CTCLoss(blank=0, reduction="mean", zero_infinity=True); "mean" divides each
utterance's loss by clamp_min(target_length, 1) and averages over the batch.
PINNED: tests/golden/ctc_*.npz hold the reference CtcLoss module's loss/grad
(imported from the reference tree by tools/gen_golden.py).
"""
import numpy as np


def _lse(a, axis=-1):
    m = np.max(a, axis=axis, keepdims=True)
    m = np.where(np.isfinite(m), m, 0.0)
    return (m + np.log(np.sum(np.exp(a - m), axis=axis, keepdims=True))).squeeze(axis)


def ctc_loss(logits, targets, logits_length, targets_length, blank=0, reduction="mean",
             zero_infinity=True):
    """logits (B,T,V); targets (B,U) padded; returns (loss, grad_logits, per_utt_nll)."""
    logits = np.asarray(logits, dtype=np.float64)
    B, T, V = logits.shape
    lp = logits - _lse(logits, -1)[..., None]
    nll = np.zeros(B)
    grad = np.zeros_like(logits)
    for b in range(B):
        Tb, Ub = int(logits_length[b]), int(targets_length[b])
        ext = np.full(2 * Ub + 1, blank, dtype=np.int64)
        ext[1::2] = np.asarray(targets[b][:Ub])
        S = ext.shape[0]
        if Tb == 0:
            nll[b] = 0.0 if Ub == 0 else np.inf
            continue
        skip = np.zeros(S, dtype=bool)
        skip[2:] = (ext[2:] != ext[:-2]) & (np.arange(2, S) % 2 == 1)
        al = np.full((Tb, S), -np.inf)
        al[0, 0] = lp[b, 0, ext[0]]
        if S > 1:
            al[0, 1] = lp[b, 0, ext[1]]
        for t in range(1, Tb):
            a = al[t - 1]
            b1 = np.concatenate([[-np.inf], a[:-1]])
            b2 = np.where(skip, np.concatenate([[-np.inf, -np.inf], a])[:S], -np.inf)
            al[t] = np.logaddexp(np.logaddexp(a, b1), b2) + lp[b, t, ext]
        ll = np.logaddexp(al[Tb - 1, S - 1], al[Tb - 1, S - 2] if S > 1 else -np.inf)
        nll[b] = -ll
        if not np.isfinite(ll):
            continue
        be = np.full((Tb, S), -np.inf)
        be[Tb - 1, S - 1] = lp[b, Tb - 1, ext[S - 1]]
        if S > 1:
            be[Tb - 1, S - 2] = lp[b, Tb - 1, ext[S - 2]]
        skipf = np.zeros(S, dtype=bool)
        skipf[:max(S - 2, 0)] = skip[2:]
        for t in range(Tb - 2, -1, -1):
            c = be[t + 1]
            c1 = np.concatenate([c[1:], [-np.inf]])
            c2 = np.where(skipf, np.concatenate([c, [-np.inf, -np.inf]])[2:], -np.inf)
            be[t] = np.logaddexp(np.logaddexp(c, c1), c2) + lp[b, t, ext]
        gamma = np.exp(al + be - lp[b, :Tb][:, ext] + nll[b])        # (Tb,S)
        occ = np.zeros((Tb, V))
        np.add.at(occ, (np.arange(Tb)[:, None].repeat(S, 1), ext[None, :].repeat(Tb, 0)), gamma)
        grad[b, :Tb] = np.exp(lp[b, :Tb]) - occ
    per = nll.copy()
    if zero_infinity:
        bad = ~np.isfinite(per)
        per[bad] = 0.0
        grad[bad] = 0.0
    tl = np.maximum(np.asarray(targets_length, dtype=np.float64), 1.0)
    if reduction == "mean":
        loss = np.mean(per / tl)
        grad = grad / (tl[:, None, None] * B)
    elif reduction == "sum":
        loss = np.sum(per)
    else:
        loss = per
    return loss, grad, per
