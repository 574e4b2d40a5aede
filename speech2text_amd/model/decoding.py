"""Validation-time greedy decoding (mirror of the reference's model/decoding.py:19-82,
157-177, 196-271): DecodingMethod / batch_search / reference_decoder / CtcGreedyDecoding /
RnntGreedyDecoding with the same constructor arguments and `decode(hidden_states[1,T,D]) -> str`.

The reference decodes one utterance at a time with Python loops over frames (and, for RNN-T,
predictor / joiner module calls per lattice move).  Here `batch_search` hands the WHOLE batch to
one HIP launch (csrc/decode.hip): per-frame argmax + repeat/blank collapse for CTC; for RNN-T
with the stateless predictor and a projection-free joiner the whole lattice walk runs on the
device, one workgroup per utterance.  Other predictor / joiner combinations keep the
module-by-module loop.  Beam / lexicon decoders are out of scope."""
import abc
from typing import List

import torch

from speech2text_amd import _native as N


class DecodingMethod(abc.ABC):
    @abc.abstractmethod
    def decode(self, hidden_states: torch.Tensor) -> str:
        pass

    def decode_batch(self, hidden_states: torch.Tensor, inputs_length: torch.Tensor) -> List[str]:
        return [self.decode(hidden_states[i:i + 1, :int(inputs_length[i]), :])
                for i in range(hidden_states.shape[0])]


def batch_search(hidden_states: torch.Tensor, inputs_length: torch.Tensor,
                 decode_session: DecodingMethod) -> List[str]:
    """hidden_states (B,T,D), inputs_length (B) -> list of decoded texts (reference :27-48)."""
    return decode_session.decode_batch(hidden_states, inputs_length)


def reference_decoder(tensor: torch.Tensor, tokenizer) -> List[str]:
    """Label rows (0-padded) -> texts (reference :157-177)."""
    refs = []
    rows = tensor.long().cpu()
    for b in range(rows.shape[0]):
        ids = []
        for u in rows[b].tolist():
            if u == 0:
                break
            ids.append(u)
        refs.append(tokenizer.decode(torch.tensor(ids, dtype=torch.int64)))
    return refs


def ctc_greedy_tokens(logits: torch.Tensor, lengths: torch.Tensor, blank: int = 0):
    """(B,T,V) device logits -> (tokens (B,T) int64, out_len (B) int64).  HIP: decode.hip."""
    if not logits.is_cuda:
        raise RuntimeError("speech2text_amd decoding runs on the GPU only (no CPU fallback)")
    logits = logits.contiguous().float()
    B, T, V = logits.shape
    lengths = lengths.to(device=logits.device, dtype=torch.int64).contiguous()
    tokens = torch.zeros((B, T), dtype=torch.int64, device=logits.device)
    out_len = torch.zeros((B,), dtype=torch.int64, device=logits.device)
    N.check(N.lib().s2t_ctc_greedy(N.fp(logits), N.lp(lengths), B, T, V, int(blank),
                                   N.lp(tokens), N.lp(out_len), N.stream()), "s2t_ctc_greedy")
    return tokens, out_len


def _to_texts(tokens, out_len, tokenizer):
    tok, n = tokens.cpu(), out_len.cpu().tolist()
    return [tokenizer.decode(tok[b, :n[b]]) for b in range(tok.shape[0])]


class CtcGreedyDecoding(DecodingMethod):
    def __init__(self, tokenizer, dummy=-1) -> None:
        self._tokenizer = tokenizer

    def decode_batch(self, hidden_states, inputs_length):
        assert hidden_states.shape[-1] == len(self._tokenizer.labels)
        return _to_texts(*ctc_greedy_tokens(hidden_states, inputs_length), self._tokenizer)

    def decode(self, hidden_states: torch.Tensor) -> str:
        assert hidden_states.shape[0] == 1, "Support BatchSize = 1 only."
        n = torch.tensor([hidden_states.shape[1]], dtype=torch.int64)
        return self.decode_batch(hidden_states, n)[0]


class RnntGreedyDecoding(DecodingMethod):
    def __init__(self, tokenizer, predictor, joiner, max_token_step=10):
        self._tokenizer = tokenizer
        self._predictor = predictor
        self._joiner = joiner
        self._max_token_step = max_token_step
        assert hasattr(self._predictor, "streaming_step") and hasattr(self._joiner, "streaming_step"), \
            "Predictor and Joiner should impl streaming_step for decoding."

    def _fusable(self):
        from speech2text_amd.model.predictor.predictor import StatelessPredictor
        p = getattr(self._predictor, "predictor", self._predictor)
        return isinstance(p, StatelessPredictor) and not self._joiner._use_out_project

    def greedy_tokens(self, hidden_states, inputs_length):
        """(B,T,D) encoder output -> (tokens (B,max_out), out_len (B)); fused device search."""
        p = getattr(self._predictor, "predictor", self._predictor)
        j = self._joiner
        am = j._enc_proj(hidden_states).contiguous().float()          # (B,T,V), one GEMM
        B, T, V = am.shape
        dev = am.device
        lengths = inputs_length.to(device=dev, dtype=torch.int64).contiguous()
        max_out = T * (self._max_token_step + 1)
        tokens = torch.zeros((B, max_out), dtype=torch.int64, device=dev)
        out_len = torch.zeros((B,), dtype=torch.int64, device=dev)
        conv_w = p._conv.weight.reshape(p._embedding_dim, p._context_size).contiguous()
        N.check(N.lib().s2t_rnnt_greedy_stateless(
            N.fp(am), N.lp(lengths), N.fp(p._embedding.weight.contiguous()), N.fp(conv_w),
            N.fp(p._output_linear.weight.contiguous()), N.fp(p._output_linear.bias.contiguous()),
            N.fp(j._pre_proj.weight.contiguous()), N.fp(j._pre_proj.bias.contiguous()), B, T, V,
            p._embedding_dim, p._output_dim, p._context_size, 0 if j._act_name == "relu" else 1,
            int(self._max_token_step), max_out, 0, N.lp(tokens), N.lp(out_len), N.stream()),
            "s2t_rnnt_greedy_stateless")
        return tokens, out_len

    @torch.no_grad()
    def decode_batch(self, hidden_states, inputs_length):
        if self._fusable():
            return _to_texts(*self.greedy_tokens(hidden_states, inputs_length), self._tokenizer)
        return super().decode_batch(hidden_states, inputs_length)

    @torch.no_grad()
    def decode(self, hidden_states: torch.Tensor) -> str:
        assert hidden_states.shape[0] == 1, "Support BatchSize = 1 only."
        if self._fusable():
            n = torch.tensor([hidden_states.shape[1]], dtype=torch.int64)
            return self.decode_batch(hidden_states, n)[0]
        # module-by-module lattice walk (reference :237-271) for LSTM predictors / out-projection
        pred_state = self._predictor.init_state()
        T = hidden_states.shape[1]
        t = 0
        cur = torch.zeros((1, 1), dtype=torch.int64, device=hidden_states.device)
        nts = 0
        pred_out, pred_state = self._predictor.streaming_step(cur, pred_state)
        out = []
        while t < T:
            logp = self._joiner.streaming_step(hidden_states[:, t:t + 1, :], pred_out)
            tok = int(logp.argmax(dim=-1))
            if tok == 0 or nts > self._max_token_step:
                t += 1
                nts = 0
            else:
                nts += 1
                cur = torch.full((1, 1), tok, dtype=torch.int64, device=hidden_states.device)
                pred_out, pred_state = self._predictor.streaming_step(cur, pred_state)
                out.append(tok)
        return self._tokenizer.decode(torch.tensor(out, dtype=torch.int64))
