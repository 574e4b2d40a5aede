"""Training entry point (reference build_task.py:32-148): same TaskFactory enum keyed by
config["task"]["type"], same `--training_config=PATH` flag (argparse here; gflags is absent),
seeds 1234, YAML load, then speech2text_amd.trainer.Trainer(**config["trainer"]).fit."""
import argparse
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


def run_task(config, batches=None, val_batches=None):
    random.seed(1234)
    np.random.seed(1234)
    torch.manual_seed(1234)
    task = TaskFactory.get(config["task"]["type"])(config)
    trainer = Trainer(**config["trainer"])
    if batches is None:
        raise NotImplementedError(
            "dataset loading is outside the accelerated path (SURVEY.md 8f); pass an iterable "
            "of batch dicts following dataset/utils.py:182-202 (or carrying 'pcm'/'pcm_length')")
    trainer.fit(task, batches, val_batches=val_batches)     # validation: Trainer.validate / val_history
    return task, trainer


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--training_config", default="config/training/zipformer_stateless_pruned_rnnt.yaml")
    args = ap.parse_args()
    with open(args.training_config) as f:
        cfg = yaml.safe_load(f)
    run_task(cfg)
