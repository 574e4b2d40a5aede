"""Metrics for evaluation (mirror of the reference's model/utils.py:23-136): word / character
error rate over decoded texts and the AsrMetric callable the tasks' validation_step uses."""
import dataclasses
from typing import List, Tuple

import torch

from speech2text_amd.model.decoding import (CtcGreedyDecoding, RnntGreedyDecoding, batch_search,
                                            reference_decoder)


def _levenshtein(a: List, b: List) -> int:
    """Edit distance with O(min(n, m)) memory (two rolling rows)."""
    if len(a) > len(b):
        a, b = b, a
    row = list(range(len(a) + 1))
    for i, y in enumerate(b, 1):
        prev, row = row, [i] + [0] * len(a)
        for j, x in enumerate(a, 1):
            row[j] = min(prev[j] + 1, row[j - 1] + 1, prev[j - 1] + (x != y))
    return row[len(a)]


def word_error_rate(hypotheses: List[str], references: List[str], show_on_screen=True,
                    use_cer=False) -> float:
    if len(hypotheses) != len(references):
        raise ValueError("In word error rate calculation, hypotheses and references lists must "
                         "have the same number of elements. But I got:{0} and {1} "
                         "correspondingly".format(len(hypotheses), len(references)))
    scores = words = 0
    for h, r in zip(hypotheses, references):
        hl, rl = (list(h), list(r)) if use_cer else (h.split(), r.split())
        words += len(rl)
        scores += _levenshtein(hl, rl)
    return 1.0 * scores / words if words != 0 else float("inf")


@dataclasses.dataclass
class AsrMetricConfig:
    decode_method: str = "ctc_greedy_search"
    max_token_step: int = 5


class AsrMetric(object):
    def __init__(self, tokenizer, config: AsrMetricConfig, predictor=None, joiner=None):
        self._tokenizer = tokenizer
        if config.decode_method == "ctc_greedy_search":
            self._decode_sess = CtcGreedyDecoding(tokenizer=tokenizer)
        elif config.decode_method == "rnnt_greedy_search":
            self._decode_sess = RnntGreedyDecoding(tokenizer=tokenizer, predictor=predictor,
                                                   joiner=joiner,
                                                   max_token_step=config.max_token_step)
        else:
            raise NotImplementedError(config.decode_method)

    def __call__(self, hidden_states, inputs_length, ground_truth):
        references = reference_decoder(ground_truth, self._tokenizer)
        hypotheses = batch_search(hidden_states, inputs_length, self._decode_sess)
        return word_error_rate(hypotheses=hypotheses, references=references)


@dataclasses.dataclass
class SslMetricConfig:
    top_ks: Tuple[int] = (1, 5)


class SslMetric(object):
    """Top-k accuracy of the SSL task over the masked label positions (reference
    model/utils.py:139-191): a position counts when its label is among the k largest logits;
    the denominator is the number of masked positions (+1e-7)."""

    def __init__(self, config: SslMetricConfig):
        self._top_ks = tuple(config.top_ks)

    @staticmethod
    def _accuracy(logits, labels, masked_dim, top_k):
        top = logits.topk(top_k, dim=-1, largest=True, sorted=True).indices       # (B,T,k)
        keep = masked_dim.to(torch.bool)
        hit = (top == labels.unsqueeze(-1)).any(dim=-1) & keep
        # (a masked-out position never matches in the reference either: its candidates are set
        # to -1 and its label to 0)
        return hit.sum().float() / (masked_dim.sum() + 1e-7)

    def __call__(self, logits, labels, masked_dim):
        return {"top_{}_acc".format(k): self._accuracy(logits, labels, masked_dim, k)
                for k in self._top_ks}
