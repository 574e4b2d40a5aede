"""One tiny pruned-RNN-T training step on the GPU (called from __graft_entry__.smoke)."""
import random

import torch


def run(dev):
    import bench
    from speech2text_amd.build_task import TaskFactory
    from speech2text_amd.trainer import Trainer
    cfg = bench.c3_config(64)
    cfg["encoder"]["config"].update({"downsampling_factor": [1, 2], "num_encoder_layers": [1, 1],
                                     "feedforward_dim": [96, 128], "encoder_dim": [48, 64],
                                     "encoder_unmasked_dim": [32, 48], "num_heads": [4, 4],
                                     "query_head_dim": 8, "value_head_dim": 4, "pos_dim": 16,
                                     "cnn_module_kernel": [15, 7]})
    cfg["predictor"]["config"].update({"output_dim": 64, "symbol_embedding_dim": 32})
    cfg["joiner"].update({"input_dim": 64})
    random.seed(0)
    torch.manual_seed(0)
    task = TaskFactory.get("Pruned_Rnnt")(cfg)
    tr = Trainer(**cfg["trainer"]).setup(task, dev)
    task.train()
    batch = bench.make_batch(0, 2, 2.0, 5, 64, dev)
    l0 = float(tr.training_step(batch, 0))
    l1 = float(tr.training_step(batch, 1))
    assert l0 == l0 and l1 == l1, "NaN loss"
    print(f"smoke train step ok: loss {l0:.4f} -> {l1:.4f}")
