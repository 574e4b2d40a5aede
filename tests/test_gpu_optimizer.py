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


@pytest.mark.parametrize("name", ["AdamW", "Adam"])
def test_flat_adam_follows_torch(dev, name):
    """The fused flat Adam / AdamW (3 launches: chunk sums, clip factor, update + zero_grad)
    against torch.optim's own step + clip_grad_norm_ on the same gradients, two param groups,
    and a state_dict round trip in torch's layout."""
    from speech2text_amd.flat import FlatStore
    from speech2text_amd.optimizer.optim_setup import OptimSetup
    torch.manual_seed(0)
    shapes = [(300, 17), (33,), (64, 8, 3), (5,), (1000, 9)]
    ps = [torch.nn.Parameter(torch.randn(s, device=dev)) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    Opt, _ = OptimSetup({"optimizer": {"type": name}, "lr_scheduler": {"type": "Warmup"}})
    groups = lambda q: [{"params": q[:2], "lr": 3e-3, "weight_decay": 0.01},     # noqa: E731
                        {"params": q[2:], "lr": 1e-3, "weight_decay": 0.1}]
    FlatStore(ps)
    opt = Opt(groups(ps), betas=(0.9, 0.98), eps=1e-8)
    opt.pre_clip = 2.0
    opt.zero_grad_in_step = True
    topt = getattr(torch.optim, name)(groups(ref), betas=(0.9, 0.98), eps=1e-8)
    g = torch.Generator().manual_seed(1)
    for it in range(12):
        grads = [torch.randn(s, generator=g) * (4.0 if it % 5 == 2 else 0.3) for s in shapes]
        for p, r, gr in zip(ps, ref, grads):
            p.grad.copy_(gr.to(dev))
            r.grad = gr.to(dev).clone()
        torch.nn.utils.clip_grad_norm_(ref, 2.0)
        topt.step()
        opt.step()
        assert opt._flat not in (None, False), "the fused path did not run"
        if it == 5:                                       # checkpoint round trip mid-way
            sd = opt.state_dict()
            assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
            opt2 = Opt(groups(ps), betas=(0.9, 0.98), eps=1e-8)
            opt2.pre_clip, opt2.zero_grad_in_step = 2.0, True
            opt2.load_state_dict(sd)
            opt = opt2
    for p, r in zip(ps, ref):
        assert float(p.grad.abs().sum()) == 0.0            # zeroed inside the step
        np.testing.assert_allclose(p.detach().cpu().numpy(), r.detach().cpu().numpy(), atol=2e-6,
                                   rtol=2e-5)
