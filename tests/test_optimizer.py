"""CPU: flat ScaledAdam + Eden against a 30-step trajectory of the reference optimizer."""
import os

import numpy as np
import torch

from speech2text_amd.optimizer.optim_setup import Eden, OptimSetup
from speech2text_amd.optimizer.scaled_adam import ScaledAdam


def test_scaled_adam_and_eden_follow_reference_trajectory(golden_dir):
    g = np.load(os.path.join(golden_dir, "scaledadam_ref.npz"))
    ps = [torch.nn.Parameter(torch.from_numpy(g[f"init{i}"].copy())) for i in range(5)]
    Opt, Sched = OptimSetup({"optimizer": {"type": "ScaledAdam"}, "lr_scheduler": {"type": "Eden"}})
    assert Opt is ScaledAdam and Sched is Eden
    opt = Opt(ps, lr=0.045, clipping_scale=2.0, clipping_update_period=6)
    sched = Sched(opt, lr_batches=10, warmup_batches=4)
    lrs = []
    for it in range(30):
        for i, p in enumerate(ps):
            gr = torch.from_numpy(g[f"grad{it}_{i}"])
            if p.grad is None:
                p.grad = gr.clone()
            else:
                p.grad.copy_(gr)
        opt.step()
        sched.step()
        lrs.append(opt.param_groups[0]["lr"])
        if it in (0, 9, 29):
            for i, p in enumerate(ps):
                np.testing.assert_allclose(p.detach().numpy(), g[f"p{it}_{i}"], atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(lrs, g["lrs"], rtol=1e-12)


def _run(ps, opt, sched, g, lo, hi):
    for it in range(lo, hi):
        for i, p in enumerate(ps):
            gr = torch.from_numpy(g[f"grad{it}_{i}"])
            if p.grad is None:
                p.grad = gr.clone()
            else:
                p.grad.copy_(gr)
        opt.step()
        sched.step()


def test_scaled_adam_state_dict_interchanges_with_reference(golden_dir):
    """Checkpoint interchange: after 15 steps our state_dict has the reference's keys
    and values (per batch of same-shaped tensors, optimizer/scaled_adam.py:30-109), and the
    REFERENCE's step-15 state loaded into a fresh optimizer continues on its trajectory."""
    g = np.load(os.path.join(golden_dir, "scaledadam_ref.npz"))
    r = np.load(os.path.join(golden_dir, "scaledadam_state_ref.npz"))
    ps = [torch.nn.Parameter(torch.from_numpy(g[f"init{i}"].copy())) for i in range(5)]
    opt = ScaledAdam(ps, lr=0.045, clipping_scale=2.0, clipping_update_period=6)
    sched = Eden(opt, lr_batches=10, warmup_batches=4)
    _run(ps, opt, sched, g, 0, 15)
    sd = opt.state_dict()
    ours = {f"{i}:{k}" for i, e in sd["state"].items() for k in e}
    assert ours == set(r["keys"].tolist())
    for key in r["keys"].tolist():
        idx, k = key.split(":")
        v = sd["state"][int(idx)][k]
        v = v.numpy() if torch.is_tensor(v) else np.asarray(v)
        assert v.shape == r[f"state{idx}_{k}"].shape, key
        np.testing.assert_allclose(v, r[f"state{idx}_{k}"], atol=3e-6, rtol=2e-5, err_msg=key)
    # resume from the REFERENCE's state
    ps2 = [torch.nn.Parameter(torch.from_numpy(r[f"p14_{i}"].copy())) for i in range(5)]
    opt2 = ScaledAdam(ps2, lr=0.045, clipping_scale=2.0, clipping_update_period=6)
    state = {}
    for key in r["keys"].tolist():
        idx, k = key.split(":")
        a = r[f"state{idx}_{k}"]
        state.setdefault(int(idx), {})[k] = torch.from_numpy(a.copy()) if a.ndim or k not in (
            "step", "num_clipped", "model_norm_threshold") else a.item()
    groups = sd["param_groups"]
    groups[0]["lr"] = float(r["lr"])
    # order of a Lightning resume: build optimizer + scheduler, then restore both states (the
    # current lr travels in the optimizer's param_groups)
    sched2 = Eden(opt2, lr_batches=10, warmup_batches=4)
    opt2.load_state_dict({"state": state, "param_groups": groups})
    sched2.load_state_dict(sched.state_dict())
    _run(ps2, opt2, sched2, g, 15, 30)
    for i, p in enumerate(ps2):
        np.testing.assert_allclose(p.detach().numpy(), g[f"p29_{i}"], atol=3e-6, rtol=2e-5)
