"""Oracle (test infrastructure only): plain-Python restatement of the reference's greedy
decoders and WER -- model/decoding.py:51-82 (CTC greedy), :196-271 (RNN-T greedy, written
functionally over the stateless predictor / joiner parameters), model/utils.py:23-89 (WER).

PINNED by the reference's own known-answer tests: model/decoding_test.py:19-116 ("abc" / "a"
from the literal 8x6 matrix) and model/utils_test.py:19-57 (WER 0.5 / 1.0)."""
import numpy as np
import torch
import torch.nn.functional as F


def ctc_greedy(logits, length, blank=0):
    ids = np.asarray(logits)[:length].argmax(axis=-1).tolist()
    out, prev = [], blank
    for p in ids:
        if (p != prev or prev == blank) and p != blank:
            out.append(p)
        prev = p
    return out


def levenshtein(a, b):
    d = np.zeros((len(a) + 1, len(b) + 1), dtype=np.int64)
    d[:, 0] = np.arange(len(a) + 1)
    d[0, :] = np.arange(len(b) + 1)
    for i in range(1, len(a) + 1):
        for j in range(1, len(b) + 1):
            d[i, j] = min(d[i - 1, j] + 1, d[i, j - 1] + 1, d[i - 1, j - 1] + (a[i - 1] != b[j - 1]))
    return int(d[len(a), len(b)])


def word_error_rate(hyps, refs, use_cer=False):
    scores = words = 0
    for h, r in zip(hyps, refs):
        hl, rl = (list(h), list(r)) if use_cer else (h.split(), r.split())
        words += len(rl)
        scores += levenshtein(hl, rl)
    return scores / words if words else float("inf")


def rnnt_greedy_stateless(sd, ppfx, jpfx, enc, length, ctx, activation="relu", max_token_step=5):
    """The reference loop with the stateless predictor's streaming_step (state = last ctx-1
    tokens) and the projection-free joiner's streaming_step (log_softmax does not move argmax)."""
    emb, conv = sd[ppfx + "_embedding.weight"], sd[ppfx + "_conv.weight"]
    lw, lb = sd[ppfx + "_output_linear.weight"], sd[ppfx + "_output_linear.bias"]

    def pred(tokens):                                   # tokens: the last `ctx` ids
        e = F.embedding(torch.tensor([tokens]), emb).transpose(1, 2)
        o = F.conv1d(e, conv, None, groups=conv.shape[0]).transpose(1, 2)
        return F.linear(o, lw, lb)                      # (1,1,D)

    act = torch.relu if activation == "relu" else torch.tanh
    state = [0] * (ctx - 1)
    cur = 0
    pred_out = pred(state + [cur])
    state = (state + [cur])[1:] if ctx > 1 else []
    out, t, nts = [], 0, 0
    while t < length:
        am = F.linear(enc[t:t + 1].unsqueeze(0), sd[jpfx + "_enc_proj.weight"], sd[jpfx + "_enc_proj.bias"])
        lm = F.linear(pred_out, sd[jpfx + "_pre_proj.weight"], sd[jpfx + "_pre_proj.bias"])
        tok = int(act(am + lm).log_softmax(-1).argmax(-1))
        if tok == 0 or nts > max_token_step:
            t += 1
            nts = 0
        else:
            nts += 1
            pred_out = pred(state + [tok])
            state = (state + [tok])[1:] if ctx > 1 else []
            out.append(tok)
    return out
