"""LSTM predictor (reference model/predictor/lstm_predictor.py:28-109 wrapping
torchaudio.models.rnnt._Predictor; torchaudio 0.13.1 not vendored, structure restated:
Embedding -> LayerNorm -> N x layer-norm LSTM (x2g/p2g gates, c_norm/g_norm) -> dropout ->
Linear -> LayerNorm; PARITY UNPINNED).  Parameter names follow torchaudio's.  Each LSTM layer is
one sequence-long HIP kernel each way (csrc/lstm.hip); the LayerNorms around the stack are the
conformer block's HIP LayerNorm."""
import dataclasses
from typing import List, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F
from speech2text_amd import conf_kernels as ck
from speech2text_amd.model.layer.scaling import Linear


@dataclasses.dataclass
class LstmPredictorConfig:
    num_symbols: int = 128
    output_dim: int = 1024
    symbol_embedding_dim: int = 512
    num_lstm_layers: int = 3
    lstm_hidden_dim: int = 512
    lstm_layer_norm: bool = True
    lstm_layer_norm_epsilon: float = 1e-3
    lstm_dropout: float = 0.3


class _CustomLSTM(nn.Module):
    def __init__(self, input_dim, hidden_dim, layer_norm=False, layer_norm_epsilon=1e-5):
        super().__init__()
        self.x2g = Linear(input_dim, 4 * hidden_dim, bias=(not layer_norm))
        self.p2g = Linear(hidden_dim, 4 * hidden_dim, bias=False)
        if layer_norm:
            self.c_norm = nn.LayerNorm(hidden_dim, eps=layer_norm_epsilon)
            self.g_norm = nn.LayerNorm(4 * hidden_dim, eps=layer_norm_epsilon)
        else:
            self.c_norm = nn.Identity()
            self.g_norm = nn.Identity()
        self.hidden_dim = hidden_dim

    def forward(self, x, state):
        """x (T,B,E) -> (hs (T,B,H), [h_T, c_T]).  The input projection of all steps is one GEMM;
        the recurrence (g_norm(x2g(x_t) + p2g(h)), gates, c_norm, h) runs as ONE kernel over the
        whole sequence, one workgroup per utterance (csrc/lstm.hip), forward and backward."""
        h0, c0 = (None, None) if state is None else state
        hs, h, c = ck.lnlstm(self.x2g(x), self.p2g.weight, self.g_norm, self.c_norm, h0, c0)
        return hs, [h, c]


class _Predictor(nn.Module):
    def __init__(self, num_symbols, output_dim, symbol_embedding_dim, num_lstm_layers,
                 lstm_hidden_dim, lstm_layer_norm=False, lstm_layer_norm_epsilon=1e-5,
                 lstm_dropout=0.0):
        super().__init__()
        self.embedding = nn.Embedding(num_symbols, symbol_embedding_dim)
        self.input_layer_norm = nn.LayerNorm(symbol_embedding_dim)
        self.lstm_layers = nn.ModuleList([
            _CustomLSTM(symbol_embedding_dim if i == 0 else lstm_hidden_dim, lstm_hidden_dim,
                        lstm_layer_norm, lstm_layer_norm_epsilon) for i in range(num_lstm_layers)])
        self.dropout = ck.Dropout(p=lstm_dropout)
        self.linear = Linear(lstm_hidden_dim, output_dim)
        self.output_layer_norm = nn.LayerNorm(output_dim)

    def forward(self, input, lengths, state=None):
        x = ck.layer_norm(self.embedding(input.permute(1, 0)), self.input_layer_norm)
        state_out = []
        for i, lstm in enumerate(self.lstm_layers):
            x, s = lstm(x, None if state is None else state[i])
            x = self.dropout(x)
            state_out.append(s)
        x = ck.layer_norm(self.linear(x), self.output_layer_norm)
        return x.permute(1, 0, 2), lengths, state_out


class LstmPredictor(nn.Module):
    def __init__(self, config: LstmPredictorConfig) -> None:
        super().__init__()
        self._sos_token = config.num_symbols - 1
        self._blank_token = 0
        self._predictor = _Predictor(**dataclasses.asdict(config))

    @property
    def sos_token(self) -> int:
        return self._sos_token

    @property
    def blank_token(self) -> int:
        return self._blank_token

    def init_state(self):
        return []

    def forward(self, input: torch.Tensor, lengths: torch.Tensor, state: List[List[torch.Tensor]]):
        x = F.pad(input.to(torch.int32), (1, 0), value=self._blank_token)     # (B, 1+U)
        return self._predictor(x, lengths, None if len(state) == 0 else state)

    @torch.no_grad()
    def streaming_step(self, input: torch.Tensor, state: List[List[torch.Tensor]]):
        """One token in, one prediction out (reference lstm_predictor.py:90-109): no left padding;
        [] = the initial state."""
        assert input.shape[0] == 1 and input.shape[1] == 1
        lengths = torch.ones(1, dtype=torch.int64, device=input.device)
        out, _, state_out = self._predictor(input.to(torch.int32), lengths,
                                            None if len(state) == 0 else state)
        return out, state_out
