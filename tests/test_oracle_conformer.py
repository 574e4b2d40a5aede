"""Conformer Subsampling (reference-authored part) pinned by reference goldens: the oracle on
the CPU, the product (channel-last convs + kernel-backed Linear) on the GPU."""
import os

import numpy as np
import pytest
import torch

from oracle import conformer as OC


def test_subsampling4_oracle_vs_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "subsampling_ref.npz"))
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd4.")}
    y, yl = OC.subsampling4({"p." + k: v for k, v in sd.items()}, "p.", torch.from_numpy(g["x"]),
                            torch.from_numpy(g["lens"]))
    np.testing.assert_allclose(y.numpy(), g["out4"], atol=1e-5)
    assert (yl.numpy() == g["len4"]).all()


@pytest.mark.gpu
def test_product_subsampling_state_dict_and_values(golden_dir, dev):
    from speech2text_amd.model.encoder.conformer import Subsampling
    g = np.load(os.path.join(golden_dir, "subsampling_ref.npz"))
    for rate in (4, 6, 8):
        m = Subsampling(80, 32, rate)
        m.load_state_dict({k[len(f"sd{rate}."):]: torch.from_numpy(g[k]) for k in g.files
                           if k.startswith(f"sd{rate}.")})
        m.to(dev)
        y, yl = m(torch.from_numpy(g["x"]).to(dev), torch.from_numpy(g["lens"]).to(dev))
        np.testing.assert_allclose(y.detach().cpu().numpy(), g[f"out{rate}"], atol=2e-5)
        assert (yl.cpu().numpy() == g[f"len{rate}"]).all()
