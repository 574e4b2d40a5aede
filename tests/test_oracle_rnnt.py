"""CPU self-checks of the (unpinned) k2 / torchaudio RNN-T restatement."""
import itertools

import numpy as np
import pytest
import torch

from oracle import k2_rnnt as K


def brute_force_score(px, py, S, T):
    """log-sum over all monotone paths from (0,0) to (S,T)."""
    tot = -np.inf
    for pos in itertools.combinations(range(S + T), S):
        s = t = 0
        lp = 0.0
        for i in range(S + T):
            if i in pos:
                lp += px[s, t]
                s += 1
            else:
                lp += py[s, t]
                t += 1
        tot = np.logaddexp(tot, lp)
    return tot


def test_mutual_information_vs_brute_force():
    rng = np.random.default_rng(0)
    B, S, T = 3, 3, 4
    px = rng.standard_normal((B, S, T + 1)).astype(np.float32)
    py = rng.standard_normal((B, S + 1, T)).astype(np.float32)
    bnd = np.array([[0, 0, 3, 4], [0, 0, 2, 4], [0, 0, 3, 2]])
    px_f = K.fix_for_boundary(torch.from_numpy(px), torch.from_numpy(bnd)).numpy()
    p, ans, gx, gy = K.mutual_information_np(px_f, py, bnd)
    for b in range(B):
        Sb, Tb = bnd[b, 2], bnd[b, 3]
        ref = brute_force_score(px_f[b], py[b], Sb, Tb)  # no symbol after the last frame
        assert abs(ans[b] - ref) < 1e-4
        # occupation counts: every path takes exactly Sb symbol arcs and Tb blank arcs
        assert abs(gx[b].sum() - Sb) < 1e-4 and abs(gy[b].sum() - Tb) < 1e-4


def test_mutual_information_grad_numeric():
    rng = np.random.default_rng(1)
    B, S, T = 1, 3, 3
    px = rng.standard_normal((B, S, T + 1))
    py = rng.standard_normal((B, S + 1, T))
    bnd = np.array([[0, 0, S, T]])
    px[:, :, T] = -np.inf
    _, ans, gx, gy = K.mutual_information_np(px, py, bnd)
    eps = 1e-6
    for (s, t) in [(0, 0), (1, 2), (2, 1)]:
        q = px.copy(); q[0, s, t] += eps
        assert abs((K.mutual_information_np(q, py, bnd)[1][0] - ans[0]) / eps - gx[0, s, t]) < 1e-4
        q = py.copy(); q[0, s, t] += eps
        assert abs((K.mutual_information_np(px, q, bnd)[1][0] - ans[0]) / eps - gy[0, s, t]) < 1e-4


def _rand_case(seed, B=3, T=12, S=5, C=9):
    g = torch.Generator().manual_seed(seed)
    am = torch.randn(B, T, C, generator=g)
    lm = torch.randn(B, S + 1, C, generator=g)
    sym = torch.randint(1, C, (B, S), generator=g)
    tl = torch.tensor([S, S - 2, S - 1][:B])
    el = torch.tensor([T, T - 3, T - 1][:B])
    return am, lm, sym, tl, el


def test_simple_loss_equals_full_loss_on_additive_logits():
    am, lm, sym, tl, el = _rand_case(0)
    B = am.shape[0]
    bnd = torch.zeros(B, 4, dtype=torch.int64); bnd[:, 2] = tl; bnd[:, 3] = el
    simple = K.rnnt_loss_smoothed(lm, am, sym, 0, bnd, return_grad=False)
    full = K.rnnt_loss_full(am.unsqueeze(2) + lm.unsqueeze(1), sym, el, tl)
    assert abs(simple.item() - full.item()) < 1e-4


def test_pruned_equals_unpruned_when_range_covers_all():
    am, lm, sym, tl, el = _rand_case(1)
    S = sym.shape[1]
    logits, bnd, ranges, simple = K.joiner_pruned(am, lm, sym, tl, el, prune_range=S + 1)
    assert (ranges[:, :, 0] == 0).all()
    pruned = K.rnnt_loss_pruned(logits, sym, ranges, 0, bnd)
    full = K.rnnt_loss_full(torch.relu(am.unsqueeze(2) + lm.unsqueeze(1)), sym, el, tl)
    assert abs(pruned.item() - full.item()) < 1e-4


def test_prune_range_invariants():
    am, lm, sym, tl, el = _rand_case(2, B=3, T=40, S=12, C=17)
    R = 5
    _, bnd, ranges, _ = K.joiner_pruned(am, lm, sym, tl, el, prune_range=R)
    s0 = ranges[:, :, 0]
    assert (s0[:, 0] == 0).all()
    d = s0[:, 1:] - s0[:, :-1]
    assert (d >= 0).all() and (d < R).all()
    assert (ranges[:, :, -1] <= sym.shape[1]).all()
    for b in range(3):
        assert s0[b, el[b] - 1] + R - 1 >= tl[b]        # last valid frame reaches S_b


def test_rnnt_full_loss_gradcheck_vs_autograd_dp():
    # independent O(TU) alpha recursion written with torch ops (autograd gives the grad)
    torch.manual_seed(3)
    B, T, U, V = 2, 5, 3, 6
    logits = torch.randn(B, T, U + 1, V, requires_grad=True)
    tg = torch.randint(1, V, (B, U))
    tl = torch.tensor([U, U - 1]); el = torch.tensor([T, T - 1])
    loss = K.rnnt_loss_full(logits, tg, el, tl)
    loss.backward()
    g1 = logits.grad.clone()
    logits.grad = None
    lp = torch.log_softmax(logits, -1)
    tot = 0
    for b in range(B):
        Tb, Ub = int(el[b]), int(tl[b])
        al = {}
        for t in range(Tb):
            for u in range(Ub + 1):
                if t == 0 and u == 0:
                    al[(t, u)] = torch.zeros(())
                    continue
                terms = []
                if t > 0:
                    terms.append(al[(t - 1, u)] + lp[b, t - 1, u, 0])
                if u > 0:
                    terms.append(al[(t, u - 1)] + lp[b, t, u - 1, tg[b, u - 1]])
                al[(t, u)] = torch.logsumexp(torch.stack(terms), 0)
        tot = tot - (al[(Tb - 1, Ub)] + lp[b, Tb - 1, Ub, 0])
    (tot / B).backward()
    assert abs(loss.item() - (tot / B).item()) < 1e-5
    assert torch.allclose(g1, logits.grad, atol=1e-5)
