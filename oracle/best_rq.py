"""Oracle: BEST-RQ labels, numpy restatement of model/ssl/best_rq.py:168-217 (frame stacking
by two unfold(1,3,2)) and :259-294 (_make_label, cosine / euclidean basis), float64
arithmetic on the fp32 inputs, first index on ties.
PINNED: tests/golden/bestrq_ref.npz holds labels produced by the reference BestRQLayer
(fp32 torch-CPU); the test reports any near-tie disagreement explicitly."""
import numpy as np


def label_lengths(length):
    length = np.asarray(length)
    for _ in range(2):
        length = (length - 3) // 2 + 1
    return length


def stack_frames(feats):
    """(B,T,F) -> (B,T2,F*9) with stacked[b,t2,d*9+k1*3+k2] = feats[b,4*t2+2*k2+k1,d]."""
    B, T, F = feats.shape
    T1 = (T - 3) // 2 + 1
    T2 = (T1 - 3) // 2 + 1
    out = np.empty((B, T2, F, 3, 3), feats.dtype)
    for k1 in range(3):
        for k2 in range(3):
            idx = 4 * np.arange(T2) + 2 * k2 + k1
            out[:, :, :, k1, k2] = feats[:, idx, :]
    return out.reshape(B, T2, F * 9)


def make_labels(feats, projector, codebooks):
    """returns (ncb, B, T2) int64 labels in 1..K"""
    st = stack_frames(np.asarray(feats, np.float32)).astype(np.float64)
    tg = st @ np.asarray(projector, np.float32).astype(np.float64)
    tg = tg / np.maximum(np.linalg.norm(tg, axis=-1, keepdims=True), 1e-12)
    labs = []
    for cb in codebooks:
        c = np.asarray(cb, np.float32).astype(np.float64)
        c = c / np.maximum(np.linalg.norm(c, axis=-1, keepdims=True), 1e-12)
        labs.append(np.argmax(tg @ c.T, axis=-1) + 1)
    return np.stack(labs, 0).astype(np.int64)
