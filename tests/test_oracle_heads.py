"""CPU: oracle/heads.py (predictor, joiner projections, CMVN, Projector, masked KL / CE, task
loss formulas) against tests/golden/heads_ref.npz = outputs of the reference classes
(tools/gen_golden.py gen_heads); and the plain-C mutual-information oracle against the numpy one."""
import os

import numpy as np
import pytest
import torch

from oracle import heads as H
from oracle import k2_rnnt as K2


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "heads_ref.npz"))


@pytest.mark.parametrize("ci", [0, 1, 2, 3])
def test_stateless_predictor_vs_reference(g, ci):
    V, D, E, ctx = (int(v) for v in g[f"pred{ci}_cfg"])
    sd = {k[len(f"pred{ci}_sd_"):]: torch.from_numpy(g[k]).requires_grad_(True)
          for k in g.files if k.startswith(f"pred{ci}_sd_")}
    lab = torch.from_numpy(g[f"pred{ci}_labels"])
    y = H.stateless_predictor(sd, "", lab, ctx)
    np.testing.assert_allclose(y.detach().numpy(), g[f"pred{ci}_out"], atol=2e-6, rtol=1e-5)
    (y * torch.from_numpy(g[f"pred{ci}_w"])).sum().backward()
    for k, v in sd.items():
        np.testing.assert_allclose(v.grad.numpy(), g[f"pred{ci}_grad_{k}"], atol=2e-5, rtol=1e-4)
    # out_state = last `context` tokens of [state | blank | labels] (reference :87-88)
    tok = np.concatenate([np.zeros((lab.shape[0], ctx), np.int64), lab.numpy()], 1)
    np.testing.assert_array_equal(g[f"pred{ci}_state"], tok[:, tok.shape[1] - ctx:])


@pytest.mark.parametrize("ci", [0, 1, 2, 3])
@pytest.mark.parametrize("kind", ["kl", "ce"])
def test_masked_ssl_losses_vs_reference(g, ci, kind):
    K, eps, scale = g[f"ssl{ci}_cfg"]
    fn = H.masked_kl_div if kind == "kl" else H.masked_ce
    for mname, mk in (("mask", torch.from_numpy(g[f"ssl{ci}_mask2d"])),
                      ("len", torch.from_numpy(g[f"ssl{ci}_lens"]))):
        lg = torch.from_numpy(g[f"ssl{ci}_logits"]).requires_grad_(True)
        loss = fn(lg, torch.from_numpy(g[f"ssl{ci}_labels"]), mk, int(K), float(scale), float(eps))
        np.testing.assert_allclose(loss.item(), g[f"ssl{ci}_{kind}_{mname}_loss"], rtol=2e-6)
        loss.backward()
        np.testing.assert_allclose(lg.grad.numpy(), g[f"ssl{ci}_{kind}_{mname}_grad"],
                                   atol=1e-7, rtol=1e-4)
    if kind == "kl":
        m = [torch.tensor(float(g[f"ssl{ci}_kl_mask_loss"])), torch.tensor(1.25)]
        t = [torch.tensor(float(g[f"ssl{ci}_kl_len_loss"])), torch.tensor(0.75)]
        sel, tot, msk = H.ssl_task_loss(m, t, "mask_loss")
        np.testing.assert_allclose([float(msk), float(tot)], g[f"ssl{ci}_task"], rtol=1e-6)
        assert float(sel) == float(msk)


def test_cmvn_projector_formulas(g):
    y = H.global_cmvn(torch.from_numpy(g["cmvn_x"]), torch.from_numpy(g["cmvn_mean"]),
                      torch.from_numpy(g["cmvn_istd"]))
    np.testing.assert_array_equal(y.numpy(), g["cmvn_y"])
    sd = {k[len("proj_sd_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("proj_sd_")}
    np.testing.assert_allclose(H.projector(sd, "", torch.from_numpy(g["proj_x"])).numpy(),
                               g["proj_y"], atol=1e-6)
    np.testing.assert_allclose(
        [H.pruned_rnnt_task_loss(3.25, 1.5, 0.5, 0.5), H.pruned_rnnt_task_loss(3.25, 1.5, 0.5, 0.5, 0.625)],
        g["formula_pruned"])
    np.testing.assert_allclose([H.hybrid_task_loss(3.25, 0.625, 0.8, 0.2)], g["formula_hybrid"])


def test_mutual_information_c_vs_numpy():
    if K2._clib() is None:
        pytest.skip("oracle/liboracle_c.so not built")
    rng = np.random.default_rng(3)
    B, S, T = 5, 9, 21
    px = (rng.standard_normal((B, S, T + 1)) - 2).astype(np.float32)
    py = (rng.standard_normal((B, S + 1, T)) - 2).astype(np.float32)
    bd = np.array([[0, 0, 9, 21], [0, 0, 4, 21], [0, 0, 9, 10], [0, 0, 1, 1], [0, 0, 0, 5]])
    for b in range(B):
        px[b, :, bd[b, 3]] = -np.inf
    a = K2.mutual_information_np(px, py, bd)
    c = K2.mutual_information_c(px, py, bd)
    for x, y in zip(a, c):
        fin = np.isfinite(x)
        assert (np.isfinite(y) == fin).all()
        np.testing.assert_allclose(y[fin], x[fin], atol=1e-6, rtol=1e-6)
