"""Optimizer / LR-scheduler factory (reference optimizer/optim_setup.py:364-385):
OptimSetup(config) -> (OptimizerClass, SchedulerClass) keyed by the YAML `type` strings."""
from enum import Enum, unique
from typing import Union

import torch
from torch.optim import Adam, AdamW
from torch.optim.lr_scheduler import CosineAnnealingLR, _LRScheduler

from speech2text_amd.optimizer.scaled_adam import ScaledAdam


class WarmupLR(_LRScheduler):
    """lr = base * warmup^0.5 * min(step^-0.5, step * warmup^-1.5)  (reference :39-80)."""

    def __init__(self, optimizer, warmup_steps: Union[int, float] = 25000, last_epoch: int = -1):
        self.warmup_steps = warmup_steps
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        step = self.last_epoch + 1
        if self.warmup_steps == 0:
            return [lr * step ** -0.5 for lr in self.base_lrs]
        return [lr * self.warmup_steps ** 0.5 * min(step ** -0.5, step * self.warmup_steps ** -1.5)
                for lr in self.base_lrs]

    def set_step(self, step: int):
        self.last_epoch = step


class Eden(_LRScheduler):
    """lr = base * ((b^2 + B^2)/B^2)^-0.5 * warmup(b)  (reference :83-135; batch-only Eden2)."""

    def __init__(self, optimizer, lr_batches: Union[int, float],
                 warmup_batches: Union[int, float] = 500.0, warmup_start: float = 0.5,
                 last_epoch: int = -1):
        self.lr_batches = lr_batches
        self.warmup_batches = warmup_batches
        assert 0.0 <= warmup_start <= 1.0
        self.warmup_start = warmup_start
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        b = self.last_epoch
        factor = ((b ** 2 + self.lr_batches ** 2) / self.lr_batches ** 2) ** -0.5
        warm = 1.0 if b >= self.warmup_batches else \
            self.warmup_start + (1.0 - self.warmup_start) * (b / self.warmup_batches)
        return [x * factor * warm for x in self.base_lrs]

    def set_step(self, step: int):
        self.last_epoch = step


@unique
class OptimizerPool(Enum):
    Adam = Adam
    AdamW = AdamW
    ScaledAdam = ScaledAdam


@unique
class LrSchedulerPool(Enum):
    Warmup = WarmupLR
    Cosine_Annealing = CosineAnnealingLR
    Eden = Eden


def OptimSetup(config):
    return (OptimizerPool[config["optimizer"]["type"]].value,
            LrSchedulerPool[config["lr_scheduler"]["type"]].value)
