"""Optimizer / LR-scheduler factory (reference optimizer/optim_setup.py:364-385):
OptimSetup(config) -> (OptimizerClass, SchedulerClass) keyed by the YAML `type` strings."""
import math
from enum import Enum, unique
from typing import Union

import torch
from torch.optim.lr_scheduler import CosineAnnealingLR, _LRScheduler

from speech2text_amd.optimizer.flat_adam import FlatAdam, FlatAdamW
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


class CosineWarmupScheduler(_LRScheduler):
    """lr = base * 0.5 (1 + cos(pi e / max_iters)) * min(1, e / warmup)  (reference :20-36; as
    shipped it cannot be constructed there -- `super().init_` -- the documented formula is kept)."""

    def __init__(self, optimizer, warmup, max_iters, last_epoch: int = -1):
        self.warmup = warmup
        self.max_num_iters = max_iters
        super().__init__(optimizer, last_epoch)

    def get_lr_factor(self, epoch):
        f = 0.5 * (1 + math.cos(math.pi * epoch / self.max_num_iters))
        if epoch <= self.warmup:
            f *= epoch * 1.0 / self.warmup
        return f

    def get_lr(self):
        f = self.get_lr_factor(self.last_epoch)
        return [lr * f for lr in self.base_lrs]


class NoamHoldAnnealing(_LRScheduler):
    """Linear warm-up to the peak lr, hold, then lr * warmup^d / (step - hold)^d  (reference
    :136-362, WarmupPolicy -> WarmupHoldPolicy -> NoamHoldAnnealing; its decay branch calls a
    method declared without `self` and raises there -- the documented formula is implemented)."""

    def __init__(self, optimizer, *, max_steps, warmup_steps=None, warmup_ratio=None,
                 hold_steps=None, hold_ratio=None, decay_rate=0.5, min_lr=0.0, last_epoch=-1):
        assert not (warmup_steps is not None and warmup_ratio is not None)
        assert not (hold_steps is not None and hold_ratio is not None)
        self.max_steps = max_steps
        self.warmup_steps = warmup_steps if warmup_steps is not None else \
            (int(warmup_ratio * max_steps) if warmup_ratio is not None else 0)
        if hold_steps is not None:
            self.hold_steps = hold_steps + self.warmup_steps
        elif hold_ratio is not None:
            self.hold_steps = int(hold_ratio * max_steps) + self.warmup_steps
        else:
            self.hold_steps = 0
        self.decay_rate = decay_rate
        self.min_lr = min_lr
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        step = self.last_epoch
        if step <= self.warmup_steps and self.warmup_steps > 0:
            return [lr * (step + 1) / (self.warmup_steps + 1) for lr in self.base_lrs]
        if self.warmup_steps <= step < self.hold_steps:
            return list(self.base_lrs)
        if step > self.max_steps:
            return [self.min_lr for _ in self.base_lrs]
        if not self.warmup_steps:
            raise ValueError("Noam scheduler cannot be used without warmup steps")
        hold = self.hold_steps - self.warmup_steps if self.hold_steps > 0 else 0
        tw = max(1, self.warmup_steps ** self.decay_rate)
        th = max(1, (step - hold) ** self.decay_rate)
        return [max(lr * tw / th, self.min_lr) for lr in self.base_lrs]


@unique
class OptimizerPool(Enum):
    Adam = FlatAdam       # torch.optim.Adam / AdamW semantics, fused over the flat buffers on the GPU
    AdamW = FlatAdamW
    ScaledAdam = ScaledAdam


@unique
class LrSchedulerPool(Enum):
    Warmup = WarmupLR
    Cosine_Annealing = CosineAnnealingLR
    Cosine_Warmup = CosineWarmupScheduler
    Noam_Hold_Annealing = NoamHoldAnnealing
    Eden = Eden


def OptimSetup(config):
    return (OptimizerPool[config["optimizer"]["type"]].value,
            LrSchedulerPool[config["lr_scheduler"]["type"]].value)
