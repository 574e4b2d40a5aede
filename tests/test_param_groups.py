"""CPU: optimizer param groups of every task cover each trainable parameter exactly once
(reference ctc_task.py:201-213, rnnt_task.py:596-630), for the shipped YAMLs when the reference
tree is present, and a multi-group ScaledAdam keeps ONE flat store whose groups are ranges."""
import copy
import glob
import os

import pytest
import torch
import yaml

import bench
from speech2text_amd.build_task import TaskFactory

REF_CFG = "/root/reference/config/training"
IN_SCOPE = {"CTC", "Rnnt", "CTC_Hybrid_Rnnt", "Pruned_Rnnt", "SSL"}


def _check_groups(task):
    opt = task.configure_optimizers()["optimizer"]
    seen = {}
    for gi, g in enumerate(opt.param_groups):
        for p in g["params"]:
            assert id(p) not in seen, "parameter listed in two groups"
            seen[id(p)] = gi
    missing = [n for n, p in task.named_parameters() if p.requires_grad and id(p) not in seen]
    assert not missing, f"parameters outside every optimizer group: {missing[:5]}"
    return opt


def test_pruned_task_with_ctc_head_has_five_groups():
    cfg = bench.c3_config(64)
    cfg["encoder"]["config"].update({"downsampling_factor": [1, 2], "num_encoder_layers": [1, 1],
                                     "feedforward_dim": [96, 128], "encoder_dim": [48, 64],
                                     "encoder_unmasked_dim": [32, 48], "num_heads": [4, 4],
                                     "query_head_dim": 8, "value_head_dim": 4, "pos_dim": 16,
                                     "cnn_module_kernel": [15, 7]})
    cfg["predictor"]["config"].update({"output_dim": 64, "symbol_embedding_dim": 32})
    cfg["joiner"].update({"input_dim": 64})
    cfg["loss"].update({"enable_ctc": True, "ctc_config": {"blank_label": 0, "reduction": "mean"}})
    cfg["ctc_projector"] = {"model": "Projector", "config": {"input_dim": 64, "output_dim": 64,
                                                            "dropout_p": 0.0}}
    cfg["optim_setup"]["seperate_lr"] = {"apply": True, "config": {
        "encoder_lr": 0.04, "decoder_lr": 0.03, "predictor_lr": 0.02, "joiner_lr": 0.01,
        "ctc_projector_lr": 0.005}}
    task = TaskFactory.get("Pruned_Rnnt")(cfg)
    opt = _check_groups(task)
    names = [g["name"] for g in opt.param_groups if g["params"]]
    assert "ctc_projector_lr" in names and "encoder_lr" in names
    assert [g.get("initial_lr", g["lr"]) for g in opt.param_groups
            if g.get("name") == "ctc_projector_lr"] == [0.005]


def test_ctc_task_seperate_lr_groups():
    cfg = bench.c2_config(32, layers=1)
    cfg["encoder"]["config"].update({"input_dim": 32, "ffn_dim": 64, "output_dim": 32})
    cfg["decoder"]["config"].update({"input_dim": 32})
    cfg["optim_setup"]["seperate_lr"] = {"apply": True, "config": {"encoder_lr": 1e-3,
                                                                   "decoder_lr": 2e-3}}
    opt = _check_groups(TaskFactory.get("CTC")(cfg))
    assert [g["name"] for g in opt.param_groups] == ["encoder_lr", "decoder_lr"]


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="reference tree not present")
def test_shipped_yamls_build_and_group_every_parameter(monkeypatch):
    # the YAMLs name their tokenizer files relative to the reference's root (its tests run from there)
    monkeypatch.chdir(os.path.dirname(os.path.dirname(REF_CFG)))
    n = 0
    for path in sorted(glob.glob(os.path.join(REF_CFG, "*.yaml"))):
        cfg = yaml.safe_load(open(path))
        if cfg["task"]["type"] not in IN_SCOPE or cfg["encoder"]["model"] not in ("Conformer",
                                                                                  "Zipformer"):
            continue
        cfg = copy.deepcopy(cfg)
        task = TaskFactory.get(cfg["task"]["type"])(cfg)
        _check_groups(task)
        assert task._metric is not None and callable(getattr(task, "validation_step"))
        n += 1
    assert n >= 6


def test_multi_group_scaled_adam_shares_one_store():
    from speech2text_amd.flat import get_store
    from speech2text_amd.optimizer.scaled_adam import ScaledAdam
    torch.manual_seed(0)
    a = [torch.nn.Parameter(torch.randn(7, 5)), torch.nn.Parameter(torch.randn(3))]
    b = [torch.nn.Parameter(torch.randn(11)), torch.nn.Parameter(torch.randn(()))]
    store = get_store(a + b)                                   # what Trainer.setup does
    opt = ScaledAdam([{"params": a, "lr": 0.04}, {"params": b, "lr": 0.01}], clipping_scale=2.0)
    norms = []
    for it in range(6):
        for p in a + b:
            p.grad.copy_(torch.randn_like(p))
        opt.step()
        assert opt.store is store, "the optimizer must reuse the model's store"
        norms.append(float(store.g().norm()))
        opt.zero_grad()
        assert float(store.g().abs().sum()) == 0.0            # grads really are zeroed
        for p in a + b:
            assert p.grad.data_ptr() >= store.flat_g.data_ptr()
    assert all(n > 0 for n in norms)
    with pytest.raises(RuntimeError):
        from speech2text_amd.flat import FlatStore
        FlatStore(a)                                          # already owned by `store`
