"""GPU: conformer encoder (fused GLU+depthwise HIP kernel, time-major) vs the oracle's plain
formulation, and one training step of every task in the TaskFactory."""
import random

import numpy as np
import pytest
import torch

from oracle import conformer as OC

pytestmark = pytest.mark.gpu


def test_conformer_vs_oracle(dev):
    from speech2text_amd.model.encoder.conformer import Conformer, ConformerConfig
    torch.manual_seed(0)
    cfg = ConformerConfig(input_dim=64, num_heads=4, ffn_dim=128, num_layers=2,
                          depthwise_conv_kernel_size=15, dropout=0.0, output_dim=40)
    m = Conformer(cfg).to(dev)
    x = torch.randn(3, 203, 80)
    lens = torch.tensor([203, 150, 99])
    pnames = {n for n, _ in m.named_parameters()}
    sd = {k: v.detach().cpu().clone().requires_grad_(k in pnames) for k, v in m.state_dict().items()}
    for training in (False, True):
        m.train(training)
        xc = x.clone().requires_grad_(True)
        yo, lo = OC.conformer_forward(sd, xc, lens, 2, 4, training=training)
        xg = x.to(dev).requires_grad_(True)
        y, l = m(xg, lens.to(dev))
        assert torch.equal(l.cpu(), lo)
        np.testing.assert_allclose(y.detach().cpu().numpy(), yo.detach().numpy(), atol=2e-4, rtol=2e-3)
        if training:
            w = torch.randn_like(yo)
            (yo * w).sum().backward()
            (y * w.to(dev)).sum().backward()
            np.testing.assert_allclose(xg.grad.cpu().numpy(), xc.grad.numpy(), atol=2e-4, rtol=1e-2)
            for n, p in m.named_parameters():
                ref = sd[n].grad.numpy()
                # (a bias in front of BatchNorm has zero true gradient: pure rounding noise)
                assert np.abs(p.grad.cpu().numpy() - ref).max() <= 1e-2 * np.abs(ref).max() + 3e-4, n


def _base_cfg(feat_type="fbank"):
    return {"dataset": {"feat_type": feat_type,
                        "feat_config": {"num_mel_bins": 80, "frame_length": 25, "frame_shift": 10,
                                        "dither": 0.0} if feat_type == "fbank" else
                        {"num_mel_bins": 80, "snip_edges": True}},
            "optim_setup": {"seperate_lr": {"apply": False},
                            "optimizer": {"type": "ScaledAdam", "config": {"lr": 0.045, "clipping_scale": 2.0}},
                            "lr_scheduler": {"type": "Eden", "config": {"lr_batches": 7000},
                                             "step_config": {"interval": "step", "frequency": 1}}},
            "trainer": {"accelerator": "gpu", "devices": 1, "strategy": "ddp", "precision": "32-true",
                        "max_epochs": 1, "accumulate_grad_batches": 2, "gradient_clip_val": 5.0,
                        "gradient_clip_algorithm": "norm"}}


_CONF = {"model": "Conformer", "config": {"bn_cmvn": False, "feats_dim": 80, "subsampling_rate": 4,
                                          "input_dim": 64, "num_heads": 4, "ffn_dim": 128,
                                          "num_layers": 2, "depthwise_conv_kernel_size": 15,
                                          "dropout": 0.1, "output_dim": 64}}


def _pcm_batch(dev, B=3, sec=2.0, U=6, V=32):
    g = torch.Generator().manual_seed(1)
    n = int(sec * 16000)
    return {"pcm": (torch.randn(B, n, generator=g) * 0.1).to(dev),
            "pcm_length": torch.tensor([n, n - 3000, n - 8000][:B]).to(dev),
            "label": torch.randint(1, V - 1, (B, U), generator=g).to(dev),
            "label_length": torch.tensor([U, U - 1, U - 3][:B]).to(dev)}


def _run(task_cls, cfg, batch, dev, steps=4):
    from speech2text_amd.trainer import Trainer
    random.seed(0); torch.manual_seed(0)
    task = task_cls(cfg)
    tr = Trainer(**cfg["trainer"]).setup(task, dev)
    task.train()
    losses = [float(tr.training_step(batch, i)) for i in range(steps)]
    assert all(np.isfinite(losses)), losses
    assert task.global_step == steps // cfg["trainer"]["accumulate_grad_batches"]
    return losses, task


def test_ctc_task_step(dev):
    from speech2text_amd.build_task import TaskFactory
    cfg = _base_cfg()
    cfg.update({"task": {"type": "CTC"}, "encoder": _CONF,
                "decoder": {"model": "Projector", "config": {"input_dim": 64, "output_dim": 32, "dropout_p": 0.1}},
                "loss": {"model": "CTC", "config": {"blank_label": 0, "reduction": "mean", "zero_infinity": True}}})
    losses, _ = _run(TaskFactory.get("CTC"), cfg, _pcm_batch(dev), dev, steps=6)
    assert losses[-1] < losses[0]


def test_hybrid_and_rnnt_task_steps(dev):
    from speech2text_amd.build_task import TaskFactory
    cfg = _base_cfg()
    cfg.update({"task": {"type": "CTC_Hybrid_Rnnt"}, "encoder": _CONF,
                "decoder": {"model": "Projector", "config": {"input_dim": 64, "output_dim": 32, "dropout_p": 0.1}},
                "predictor": {"model": "Lstm", "config": {"num_symbols": 32, "output_dim": 64,
                                                          "symbol_embedding_dim": 32, "num_lstm_layers": 2,
                                                          "lstm_hidden_dim": 48, "lstm_layer_norm": True,
                                                          "lstm_layer_norm_epsilon": 1e-3, "lstm_dropout": 0.1}},
                "joiner": {"input_dim": 64, "output_dim": 32, "inner_dim": 48, "activation": "tanh",
                           "prune_range": -1},
                "loss": {"rnnt_weight": 0.8, "ctc_weight": 0.2,
                         "rnnt_loss": {"model": "Rnnt", "config": {"blank_label": 0, "reduction": "mean"}},
                         "ctc_loss": {"model": "CTC", "config": {"blank_label": 0, "reduction": "mean"}}}})
    losses, task = _run(TaskFactory.get("CTC_Hybrid_Rnnt"), cfg, _pcm_batch(dev), dev)
    assert set(task.logged) >= {"train_loss", "train_loss/loss_rnnt", "train_loss/loss_ctc"}
    cfg2 = dict(cfg)
    cfg2["task"] = {"type": "Rnnt"}
    cfg2["decoder"] = {"model": "Identity", "config": {"dummy": -1}}
    cfg2["loss"] = {"model": "Rnnt", "config": {"blank_label": 0, "reduction": "mean"}}
    _run(TaskFactory.get("Rnnt"), cfg2, _pcm_batch(dev), dev)


def test_ssl_task_step(dev):
    from speech2text_amd.build_task import TaskFactory
    cfg = _base_cfg()
    cfg.update({"task": {"type": "SSL"}, "encoder": _CONF,
                "ssl_layer": {"model": "Best-RQ",
                              "layer_config": {"cnn_kernel_size": [3, 3], "cnn_stride": [2, 2], "feat_dim": 80,
                                               "num_codebooks": 2, "codebook_dim": 16, "codebook_size": 256,
                                               "label_basis": "cosine"},
                              "masking_config": {"mask_proportion": 0.5, "mean_span_length": 1,
                                                 "span_select_type": "static", "min_num_spans": 1,
                                                 "no_overlap": False, "min_space": 0, "seed": 1234}},
                "logits_layer": {"model": "Projector", "config": {"input_dim": 64, "output_dim": 257, "dropout_p": 0.0}},
                "loss": {"loss_select": "mask_loss", "model": "MaskedKLDiv",
                         "config": {"num_classes": 257, "scale_factor": 1.0, "label_smoothing": 0.1}}})
    g = torch.Generator().manual_seed(2)
    feats = torch.randn(3, 203, 80, generator=g).to(dev)
    batch = {"raw_feat": feats, "auged_feat": feats.clone(), "feat_length": torch.tensor([203, 180, 151]).to(dev)}
    losses, task = _run(TaskFactory.get("SSL"), cfg, batch, dev)
    assert "mask_rate" in task.logged and 0.2 < float(task.logged["mask_rate"]) < 0.8


def test_pruned_rnnt_task_step_zipformer(dev):
    from speech2text_amd.build_task import TaskFactory
    import bench
    cfg = bench.c3_config(64)
    cfg["encoder"]["config"].update({"downsampling_factor": [1, 2], "num_encoder_layers": [1, 1],
                                     "feedforward_dim": [96, 128], "encoder_dim": [48, 64],
                                     "encoder_unmasked_dim": [32, 48], "num_heads": [4, 4],
                                     "query_head_dim": 8, "value_head_dim": 4, "pos_dim": 16,
                                     "cnn_module_kernel": [15, 7], "chunk_size": [16, -1],
                                     "left_context_frames": [32, -1]})
    cfg["predictor"]["config"].update({"output_dim": 64, "symbol_embedding_dim": 32})
    cfg["joiner"].update({"input_dim": 64})
    cfg["loss"]["enable_ctc"] = True
    cfg["loss"]["ctc_config"] = {"blank_label": 0, "reduction": "mean", "zero_infinity": True}
    cfg["ctc_projector"] = {"model": "Projector", "config": {"input_dim": 64, "output_dim": 64, "dropout_p": 0.1}}
    cfg["trainer"]["accumulate_grad_batches"] = 1
    losses, task = _run(TaskFactory.get("Pruned_Rnnt"), cfg, _pcm_batch(dev, V=64), dev, steps=6)
    assert losses[-1] < losses[0]
    assert set(task.logged) >= {"train_loss", "train_loss/simple_loss", "train_loss/pruned_loss",
                                "train_loss/ctc_loss"}
