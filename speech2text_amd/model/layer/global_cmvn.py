"""Global CMVN layer (reference model/layer/global_cmvn.py:12-38): (feat - mean) * istd."""
from typing import Dict

import torch
import torch.nn as nn


class GlobalCmvnLayer(nn.Module):
    def __init__(self, config: Dict) -> None:
        super().__init__()
        if config["feat_type"] != "pcm":
            assert "num_mel_bins" in config["feat_config"]
            self._feat_dim = config["feat_config"]["num_mel_bins"]
            self.register_buffer("global_mean", torch.zeros(self._feat_dim))
            self.register_buffer("global_istd", torch.ones(self._feat_dim))
        else:
            self.register_buffer("global_mean", None)
            self.register_buffer("global_istd", None)

    def forward(self, feat: torch.Tensor) -> torch.Tensor:
        if self.global_mean is None or self.global_istd is None:
            return feat
        return (feat - self.global_mean) * self.global_istd
