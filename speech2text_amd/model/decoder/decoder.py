"""Decoder factory + heads (reference model/decoder/{decoder,identity,projector}.py)."""
import dataclasses
from typing import Tuple

import torch
import torch.nn as nn
from speech2text_amd import conf_kernels as ck
from speech2text_amd.model.layer.scaling import Linear


@dataclasses.dataclass
class IdentityConfig:
    dummy: int = -1


class Identity(nn.Module):
    def __init__(self, config: IdentityConfig):
        super().__init__()

    def forward(self, x: torch.Tensor, length: torch.Tensor):
        return x, length


@dataclasses.dataclass
class ProjectorConfig:
    input_dim: int = 512
    output_dim: int = 1000
    dropout_p: float = 0.1


class Projector(nn.Module):
    def __init__(self, config: ProjectorConfig) -> None:
        super().__init__()
        self._fc = Linear(config.input_dim, config.output_dim)
        self._dropout = ck.Dropout(p=config.dropout_p)

    def forward(self, x: torch.Tensor, length: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return self._dropout(self._fc(x)), length


class Decoder(nn.Module):
    def __init__(self, config) -> None:
        super().__init__()
        if config["model"] == "Identity":
            self.decoder = Identity(config=IdentityConfig(**config["config"]))
        elif config["model"] == "Projector":
            self.decoder = Projector(config=ProjectorConfig(**config["config"]))

    def forward(self, x: torch.Tensor, length: torch.Tensor):
        return self.decoder(x, length)
