"""GPU-side augmentation, collate and the bucket sampler against outputs of
the reference classes (tests/golden/aug_ref.npz).  CPU: oracle + sampler host logic; GPU: the
batched HIP ops with the same `random` seeds."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import augment as OA


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "aug_ref.npz"))


def test_oracle_augment_vs_reference(g):
    random.seed(4321)
    for i in range(3):
        np.testing.assert_array_equal(OA.spec_augment(g[f"sa_in{i}"]), g[f"sa_out{i}"])
    random.seed(4322)
    for i in range(3):
        np.testing.assert_allclose(OA.mix_feats(g[f"mf_src{i}"], g[f"mf_noise{i}"]), g[f"mf_out{i}"],
                                   atol=2e-5, rtol=2e-5)
    random.seed(4323)
    for i in range(3):
        np.testing.assert_allclose(OA.add_noise(g[f"an_pcm{i}"], g[f"an_noise{i}"]), g[f"an_out{i}"],
                                   atol=2e-6)


def test_bucket_sampler_vs_reference(g):
    from speech2text_amd.dataset.sampler import DynamicBucketBatchSampler
    if "bs_flat" not in g.files:
        pytest.skip("sampler golden not generated")
    durs = g["bs_durs"]

    class DS:
        lower_bound, high_bound, total_data_amount = 1.0, 15.0, float(durs.sum())

        def fetch_data_k_info(self, i, k="duration"):
            return float(durs[i])

    class SM:
        rank, num_replicas = 0, 1

        def __iter__(self):
            return iter(range(400))

        def __len__(self):
            return 400

    bs = DynamicBucketBatchSampler(SM(), DS(), num_bucket=6, min_batch_size=4, volume_threshold=60)
    it = iter(bs)
    batches = [next(it) for _ in range(25)]
    assert len(bs) == int(g["bs_len"][0])
    assert [len(b) for b in batches] == g["bs_sizes"].tolist()
    assert [i for b in batches for i in b] == g["bs_flat"].tolist()


@pytest.mark.gpu
def test_batched_augmentation_vs_reference(dev, g):
    from speech2text_amd.dataset.frontend.data_augmentation import AddNoise, MixFeats, SpecAugment
    from speech2text_amd.dataset.utils import batch as collate
    # ---- SpecAugment: the three utterances as ONE padded batch, same seed, same draw order
    feats = [torch.from_numpy(g[f"sa_in{i}"]) for i in range(3)]
    b = collate({"feat": [f.to(dev) for f in feats], "feat_length": [f.shape[0] for f in feats],
                 "label": [torch.tensor([1, 2]), torch.tensor([3]), torch.tensor([4, 5, 6])],
                 "label_length": [2, 1, 3]})
    assert b["feat"].shape == (3, 300, 80) and b["label"].tolist() == [[1, 2, 0], [3, 0, 0], [4, 5, 6]]
    for i, f in enumerate(feats):                            # collate == pad_sequence
        np.testing.assert_array_equal(b["feat"][i, :f.shape[0]].cpu().numpy(), f.numpy())
        assert float(b["feat"][i, f.shape[0]:].abs().sum()) == 0.0
    random.seed(4321)
    y = SpecAugment(2, 2, 50, 10).process_batch(b["feat"], b["feat_length"])
    for i, f in enumerate(feats):
        np.testing.assert_array_equal(y[i, :f.shape[0]].cpu().numpy(), g[f"sa_out{i}"])
    # ---- MixFeats
    random.seed(4322)
    mf = MixFeats((10, 20))
    src = [torch.from_numpy(g[f"mf_src{i}"]) for i in range(3)]
    nz = [torch.from_numpy(g[f"mf_noise{i}"]) for i in range(3)]
    sb = torch.nn.utils.rnn.pad_sequence(src, batch_first=True).to(dev)
    nb = torch.nn.utils.rnn.pad_sequence(nz, batch_first=True).to(dev)
    out = mf.process_batch(sb, [s.shape[0] for s in src], nb, [n.shape[0] for n in nz])
    for i, s in enumerate(src):
        np.testing.assert_allclose(out[i, :s.shape[0]].cpu().numpy(), g[f"mf_out{i}"], atol=3e-5,
                                   rtol=3e-5)
    # ---- AddNoise
    random.seed(4323)
    an = AddNoise(10, 50)
    pcm = [torch.from_numpy(g[f"an_pcm{i}"][0]) for i in range(3)]
    nzp = [torch.from_numpy(g[f"an_noise{i}"][0]) for i in range(3)]
    pb = torch.nn.utils.rnn.pad_sequence(pcm, batch_first=True).to(dev)
    npb = torch.nn.utils.rnn.pad_sequence(nzp, batch_first=True).to(dev)
    out = an.process_batch(pb, [p.shape[0] for p in pcm], npb, [n.shape[0] for n in nzp])
    for i, p in enumerate(pcm):
        np.testing.assert_allclose(out[i, :p.shape[0]].cpu().numpy(), g[f"an_out{i}"][0], atol=3e-6)
    # per-utterance entry points keep the reference signatures
    random.seed(4321)
    one = SpecAugment(2, 2, 50, 10).process(feats[0].to(dev))
    np.testing.assert_array_equal(one.cpu().numpy(), g["sa_out0"])
