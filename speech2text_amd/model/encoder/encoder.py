"""Encoder factory (reference model/encoder/encoder.py:15-56): YAML `model` -> implementation."""
import torch
import torch.nn as nn

from speech2text_amd.model.encoder.zipformer import Zipformer2, Zipformer2Config


class Encoder(nn.Module):
    def __init__(self, config) -> None:
        super().__init__()
        name = config["model"]
        if name == "Zipformer":
            self.encoder = Zipformer2(config=Zipformer2Config(**config["config"]))
        elif name == "Conformer":
            from speech2text_amd.model.encoder.conformer import Conformer, ConformerConfig
            self.encoder = Conformer(config=ConformerConfig(**config["config"]))
        elif name in ("Wav2Vec2", "Emformer"):
            raise NotImplementedError(
                f"{name} encoder is outside the accelerated training path (SURVEY.md section 2)")
        # unknown names leave `.encoder` unset, exactly like the reference (no else branch)

    def forward(self, x: torch.Tensor, lengths: torch.Tensor):
        return self.encoder(x, lengths)

    @torch.no_grad()
    def streaming_forward(self, x: torch.Tensor, length: torch.Tensor, **config):
        """Simulated-streaming inference interface (reference encoder.py:39-49)."""
        if hasattr(self.encoder, "streaming_forward"):
            return self.encoder.streaming_forward(x, length, **config)
        raise NotImplementedError("{} encoder does not support streaming_forward".format(
            self.encoder.__class__.__name__))
