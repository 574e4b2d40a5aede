"""CPU: BEST-RQ oracle against labels produced by the reference BestRQLayer."""
import os

import numpy as np

from oracle import best_rq as obr


def test_labels_vs_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "bestrq_ref.npz"))
    for ci in range(3):
        ncb = g[f"labels{ci}"].shape[0]
        cbs = [g[f"codebook{ci}_{j}"] for j in range(ncb)]
        lab = obr.make_labels(g[f"raw{ci}"], g[f"projector{ci}"], cbs)
        ref = g[f"labels{ci}"]
        assert lab.shape == ref.shape
        # bit-exact on every golden label (no fp32 near-tie among them)
        assert (lab == ref).all(), f"case {ci}: {int((lab != ref).sum())} of {ref.size} differ"
        assert (obr.label_lengths(g[f"length{ci}"]) <= ref.shape[2]).all()
