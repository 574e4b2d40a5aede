"""GPU: the fused multi-tensor ScaledAdam kernels (csrc/optim.hip) against (1) the reference
optimizer's own 30-step trajectory (tests/golden/scaledadam_ref.npz) and (2) the same update
written as torch ops, for several param groups, scalar tensors, odd sizes, the trainer's
grad-norm clip, threshold re-estimation (regular and the irregular k=10/20/40 steps) and
non-finite gradients."""
import os

import numpy as np
import pytest
import torch

from speech2text_amd.flat import get_store
from speech2text_amd.optimizer.optim_setup import OptimSetup
from speech2text_amd.optimizer.scaled_adam import ScaledAdam

pytestmark = pytest.mark.gpu


def test_fused_scaled_adam_follows_reference_trajectory(dev, golden_dir):
    g = np.load(os.path.join(golden_dir, "scaledadam_ref.npz"))
    ps = [torch.nn.Parameter(torch.from_numpy(g[f"init{i}"].copy()).to(dev)) for i in range(5)]
    Opt, Sched = OptimSetup({"optimizer": {"type": "ScaledAdam"}, "lr_scheduler": {"type": "Eden"}})
    opt = Opt(ps, lr=0.045, clipping_scale=2.0, clipping_update_period=6)
    sched = Sched(opt, lr_batches=10, warmup_batches=4)
    for it in range(30):
        for i, p in enumerate(ps):
            gr = torch.from_numpy(g[f"grad{it}_{i}"]).to(dev)
            if p.grad is None:
                p.grad = gr.clone()
            else:
                p.grad.copy_(gr)
        opt.step()
        sched.step()
        if it in (0, 9, 29):
            for i, p in enumerate(ps):
                np.testing.assert_allclose(p.detach().cpu().numpy(), g[f"p{it}_{i}"], atol=3e-6,
                                           rtol=2e-5)


@pytest.mark.parametrize("period,clip", [(12, 5.0), (100, None), (6, 0.5)])
def test_fused_scaled_adam_matches_torch_form(dev, period, clip):
    shapes_a = [(33, 17), (8193,), (1,), (64, 3, 3), ()]
    shapes_b = [(5,), (40, 40), (1,)]

    def make(device):
        gen = torch.Generator().manual_seed(11)
        mk = lambda shp: torch.nn.Parameter((torch.randn(shp, generator=gen) * 0.3).to(device))  # noqa
        a, b = [mk(s) for s in shapes_a], [mk(s) for s in shapes_b]
        get_store(a + b)
        opt = ScaledAdam([{"params": a, "lr": 0.04}, {"params": b, "lr": 0.01}],
                         clipping_scale=2.0, clipping_update_period=period)
        opt.pre_clip = clip
        opt.zero_grad_in_step = True
        return a + b, opt

    pc, oc = make("cpu")
    pg, og = make(dev)
    gen = torch.Generator().manual_seed(5)
    for it in range(45):
        for c, g in zip(pc, pg):
            gr = torch.randn(c.shape, generator=gen) * (10.0 if it == 23 else 1.0)
            if it == 31 and c.dim() == 2:
                gr.view(-1)[0] = float("inf")          # sanitised to zero once clipping is live
            c.grad.copy_(gr)
            g.grad.copy_(gr.to(dev))
        oc.step()
        og.step()
        if it in (0, 3, 4, 10, 12, 20, 24, 32, 44):
            for i, (c, g) in enumerate(zip(pc, pg)):
                np.testing.assert_allclose(g.detach().cpu().numpy(), c.detach().numpy(), atol=2e-6,
                                           rtol=3e-5, err_msg=f"step {it} tensor {i}")
            assert float(og.store.flat_g.abs().sum()) == 0.0     # zero_grad fused into the step
    for sc, sg in zip(oc._gstate, og._gstate):
        np.testing.assert_allclose(sg["param_rms"].cpu().numpy(), sc["param_rms"].numpy(), rtol=1e-5)
        assert int(sg["istate"][0]) == int(sc["istate"][0])
        np.testing.assert_allclose(float(sg["fstate"][0]), float(sc["fstate"][0]), rtol=1e-5)
