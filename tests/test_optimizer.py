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
