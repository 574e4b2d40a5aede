"""Training entry point (reference build_task.py:32-148): same TaskFactory enum keyed by
config["task"]["type"], same `--training_config=PATH` flag (argparse here; gflags is absent),
seeds 1234, YAML load, then speech2text_amd.trainer.Trainer(**config["trainer"]).fit."""
import argparse
import os
import random
from enum import Enum, unique

import numpy as np
import torch
import yaml

from speech2text_amd.task_factory.ctc_task import CtcTask
from speech2text_amd.task_factory.rnnt_task import CtcHybridRnnt, PrunedRnntTask, RnntTask
from speech2text_amd.task_factory.ssl_task import SslTask
from speech2text_amd.trainer import Trainer


class CifTask:
    def __init__(self, config):
        raise NotImplementedError("task CIF is outside the accelerated path (SURVEY.md 2)")


class NnlmTask:
    def __init__(self, config):
        raise NotImplementedError("task NNLM is outside the accelerated path (SURVEY.md 2)")


@unique
class TaskFactory(Enum):
    """Same keys as the reference enum (build_task.py:36-45): TaskFactory[type].value(config)."""
    CTC = CtcTask
    Rnnt = RnntTask
    CTC_Hybrid_Rnnt = CtcHybridRnnt
    Pruned_Rnnt = PrunedRnntTask
    SSL = SslTask
    CIF = CifTask
    NNLM = NnlmTask

    @classmethod
    def get(cls, name):
        return cls[name].value


def run_task(config, batches=None, val_batches=None, export_dir=None, name="task"):
    """The reference's run_task (build_task.py:44-148) around the accelerated path: seeds, task (or
    the finetune start from `finetune.base_model`: a checkpoint file, or a directory whose top-k files
    are averaged first), trainer, `resume`, the fit loop with validation passes and -- with an
    `export_dir` -- the top-k checkpoints of `callbacks.model_chkpt_config` under
    `<export_dir>/checkpoints`.  Dataset loading stays outside: `batches` /
    `val_batches` are iterables of batch dicts."""
    from speech2text_amd import checkpoint as C
    random.seed(1234)
    np.random.seed(1234)
    torch.manual_seed(1234)
    task = TaskFactory.get(config["task"]["type"])(config)
    base = (config.get("finetune") or {}).get("base_model")
    if base:
        if os.path.isdir(base):
            C.model_average(base)
            base = os.path.join(base, "averaged.chkpt")
        C.load_from_checkpoint(task, base, strict=False)
    trainer = Trainer(**config["trainer"])
    if batches is None:
        raise NotImplementedError(
            "dataset loading is outside the accelerated path (SURVEY.md 8f); pass an iterable "
            "of batch dicts following dataset/utils.py:182-202 (or carrying 'pcm'/'pcm_length')")
    trainer.setup(task)
    resumed = C.resume(trainer, config["resume"]) if config.get("resume") else None
    ck = None
    chk_cfg = (config.get("callbacks") or {}).get("model_chkpt_config")
    if export_dir is not None and chk_cfg:
        ck = {"dirpath": os.path.join(export_dir, "checkpoints"), "name": name, "config": chk_cfg,
              "resumed_best_k": C.resumed_best_k(resumed) if resumed else None}
    trainer.fit(task, batches, val_batches=val_batches, checkpoint=ck)
    return task, trainer


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--training_config", default="config/training/zipformer_stateless_pruned_rnnt.yaml")
    args = ap.parse_args()
    with open(args.training_config) as f:
        cfg = yaml.safe_load(f)
    run_task(cfg)
