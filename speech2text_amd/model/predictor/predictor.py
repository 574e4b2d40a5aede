"""Predictor factory + stateless predictor (reference model/predictor/predictor.py:17-63,
model/predictor/stateless_predictor.py:27-105): Embedding -> depthwise Conv1d(context) -> Linear
on the blank-left-padded label sequence."""
import dataclasses
from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F
from speech2text_amd import kernels as K
from speech2text_amd.model.layer.scaling import Linear


@dataclasses.dataclass
class StatelessPredictorConfig:
    num_symbols: int = 128
    output_dim: int = 1024
    symbol_embedding_dim: int = 512
    context_size: int = 5


class StatelessPredictor(nn.Module):
    def __init__(self, config: StatelessPredictorConfig) -> None:
        super().__init__()
        self._sos_token = config.num_symbols - 1
        self._blank_token = 0
        self._embedding_dim = config.symbol_embedding_dim
        self._num_symbols = config.num_symbols
        self._embedding = nn.Embedding(self._num_symbols, self._embedding_dim)
        assert config.context_size >= 1
        self._context_size = config.context_size
        self._output_dim = config.output_dim
        self._conv = nn.Conv1d(self._embedding_dim, self._embedding_dim,
                               kernel_size=self._context_size, stride=1, padding=0,
                               groups=self._embedding_dim, bias=False)
        self._output_linear = Linear(self._embedding_dim, self._output_dim)

    @property
    def sos_token(self) -> int:
        return self._sos_token

    @property
    def blank_token(self) -> int:
        return self._blank_token

    def init_state(self, batch_size: int = 1) -> torch.Tensor:
        # on the parameters' device: the reference builds it on the CPU and forward() moves it
        # (stateless_predictor.py:60-62, 86) -- a pageable host-to-device copy, i.e. a full stream
        # synchronisation in the middle of every training step's forward pass
        return torch.zeros(batch_size, self._context_size - 1, dtype=torch.int32,
                           device=self._embedding.weight.device)

    def forward(self, input: torch.Tensor, lengths: torch.Tensor,
                state: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        B = input.shape[0]
        # [state (context-1 blanks)] + [blank] + labels  -> (B, context + U)
        tokens = F.pad(input.to(torch.int32), (self._context_size, 0), value=self._blank_token)
        tokens[:, :self._context_size - 1] = state.to(input.device).repeat(B, 1)
        out_state = tokens[:, tokens.shape[1] - self._context_size:]
        return self._output_linear(self._context(tokens)), lengths, out_state

    def _context(self, tokens: torch.Tensor) -> torch.Tensor:
        """Embedding -> depthwise Conv1d over the context window: (B, L) -> (B, L-ctx+1, D).  On the
        GPU one gather kernel per pass (kernels.predictor_context); the module composition on the
        CPU (host-side tools and tests only)."""
        w = self._embedding.weight
        if (w.is_cuda and w.dtype == torch.float32 and self._context_size <= K.PRED_MAX_CONTEXT
                and self._embedding.padding_idx is None):
            return K.predictor_context(tokens, w, self._conv.weight)
        emb = self._embedding(tokens).transpose(1, 2)               # (B, D, ctx+U)
        return self._conv(emb).transpose(1, 2)                      # (B, U+1, D)

    def streaming_step(self, input: torch.Tensor, state: torch.Tensor):
        """One token in, one prediction out; state = the previous context-1 tokens
        (reference stateless_predictor.py:109-125)."""
        assert input.shape[1] == 1
        ctxed = torch.cat([state.to(input.device), input.to(state.dtype)], dim=1)
        out_state = ctxed[:, ctxed.shape[1] - self._context_size + 1:]
        return self._output_linear(self._context(ctxed)), out_state


class Predictor(nn.Module):
    def streaming_step(self, input, state):
        return self.predictor.streaming_step(input, state)

    def __init__(self, config) -> None:
        super().__init__()
        if config["model"] == "Stateless":
            self.predictor = StatelessPredictor(
                config=StatelessPredictorConfig(**config["config"]))
        elif config["model"] == "Lstm":
            from speech2text_amd.model.predictor.lstm_predictor import (LstmPredictor,
                                                                        LstmPredictorConfig)
            self.predictor = LstmPredictor(config=LstmPredictorConfig(**config["config"]))
        else:
            raise NotImplementedError

    def forward(self, input, lengths, state):
        return self.predictor(input, lengths, state)

    def init_state(self):
        return self.predictor.init_state()
