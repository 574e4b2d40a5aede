"""Tokenizers used by the validation-time metrics (mirror of the reference's
dataset/utils.py:14-178 Tokenizer / CharTokenizer / SubwordTokenizer / TokenizerSetup: same
label layout -- <blank_id> at 0, <unk> at 1 for the char tokenizer, <sos/eos> last)."""
import dataclasses
from typing import List

import torch


class Tokenizer:
    @property
    def labels(self) -> List[str]:
        raise NotImplementedError

    def _text_to_vector(self, text) -> torch.Tensor:
        unk = self.labels.index("<unk>")
        return torch.tensor([self.labels.index(c) if c in self.labels else unk for c in text],
                            dtype=torch.int64)

    def _vector_to_tokens(self, vector: torch.Tensor) -> List[str]:
        return [self.labels[i] for i in vector.tolist()]

    def export_units(self, export_filename: str) -> None:
        with open(export_filename, "w") as f:
            for i, unit in enumerate(self.labels):
                f.write("{} {}\n".format(unit, i))


@dataclasses.dataclass
class CharTokenizerConfig:
    labels: tuple = tuple("abcdefghijklmnopqrstuvwxyz' ")


class CharTokenizer(Tokenizer):
    def __init__(self, config: CharTokenizerConfig) -> None:
        self._labels = ["<blank_id>", "<unk>"] + list(config.labels) + ["<sos/eos>"]

    @property
    def labels(self):
        return self._labels

    def encode(self, text: str) -> torch.Tensor:
        return self._text_to_vector(text)

    def decode(self, vector: torch.Tensor) -> str:
        return "".join(self._vector_to_tokens(vector))

    def encode_as_tokens(self, text: str) -> List[str]:
        return [c if c in self.labels else "<unk>" for c in text]

    def decode_from_tokens(self, tokens: List[str]) -> str:
        for c in tokens:
            assert c in self.labels, "Out of vocabulary detects with '{}'".format(c)
        return "".join(tokens)


@dataclasses.dataclass
class SubwordTokenizerConfig:
    spm_model: str = None
    spm_vocab: str = None


class SubwordTokenizer(Tokenizer):
    def __init__(self, config: SubwordTokenizerConfig):
        import sentencepiece as spm
        assert config.spm_model is not None and config.spm_vocab is not None
        self._labels = ["<blank_id>"]
        with open(config.spm_vocab, "r") as f:
            for line in f:
                token = line.strip().split("\t")[0]
                if token not in ("<s>", "</s>"):
                    self._labels.append(token)
        self._labels.append("<sos/eos>")
        self._sp = spm.SentencePieceProcessor()
        self._sp.Load(config.spm_model)

    @property
    def labels(self) -> List[str]:
        return self._labels

    def encode(self, text: str) -> torch.Tensor:
        return self._text_to_vector(self._sp.EncodeAsPieces(text, emit_unk_piece=True))

    def decode(self, vector: torch.Tensor) -> str:
        return self._sp.DecodePieces(self._vector_to_tokens(vector))

    def encode_as_tokens(self, text: str) -> List[str]:
        return [t if t in self.labels else "<unk>"
                for t in self._sp.EncodeAsPieces(text, emit_unk_piece=True)]

    def decode_from_tokens(self, tokens: List[str]) -> str:
        for c in tokens:
            assert c in self.labels, "Out of vocabulary detects with '{}'".format(c)
        return self._sp.DecodePieces(tokens)


def TokenizerSetup(config) -> Tokenizer:
    if config["type"] == "char":
        return CharTokenizer(config=CharTokenizerConfig(**config["config"]))
    if config["type"] == "subword":
        return SubwordTokenizer(config=SubwordTokenizerConfig(**config["config"]))
    raise ValueError("Only 'char' and 'subword' tokenizer supported currently.")


def batch(batch, device=None):
    """Pads the variable-length entries of a batch dict and stacks them (reference
    dataset/utils.py:182-202 `batch`: pad_sequence per key on the CPU).  Here the rows are
    packed once and scattered into the zero-padded (B, Lmax, D) device tensor by one kernel."""
    from speech2text_amd import _native as N

    def pad(rows, dtype):
        dev = device or rows[0].device
        if dtype == torch.int64:                       # labels: tiny, plain torch
            return torch.nn.utils.rnn.pad_sequence([r.to(dev) for r in rows], batch_first=True,
                                                   padding_value=0)
        rows = [r.to(device=dev, dtype=torch.float32) for r in rows]
        if not rows[0].is_cuda:
            raise RuntimeError("speech2text_amd collate runs on the GPU only (no CPU fallback)")
        D = 1 if rows[0].dim() == 1 else rows[0].shape[-1]
        lens = [r.shape[0] for r in rows]
        offs = [0]
        for n in lens:
            offs.append(offs[-1] + n * D)
        packed = torch.cat([r.reshape(-1) for r in rows])
        Lmax = max(lens)
        out = torch.empty((len(rows), Lmax) + (() if rows[0].dim() == 1 else (D,)),
                          dtype=torch.float32, device=dev)
        N.check(N.lib().s2t_pad_rows(N.fp(packed), N.lp(torch.tensor(offs, device=dev)), len(rows),
                                     Lmax, D, N.fp(out), N.stream()), "s2t_pad_rows")
        return out

    if "feat" in batch and "label" in batch:
        batch["feat"] = pad(batch["feat"], torch.float32)
        batch["feat_length"] = torch.tensor(batch["feat_length"]).long()
        batch["label"] = pad([l.long() for l in batch["label"]], torch.int64)
        batch["label_length"] = torch.tensor(batch["label_length"]).long()
    elif "raw_feat" in batch and "auged_feat" in batch:
        batch["raw_feat"] = pad(batch["raw_feat"], torch.float32)
        batch["auged_feat"] = pad(batch["auged_feat"], torch.float32)
        batch["feat_length"] = torch.tensor(batch["feat_length"]).long()
    return batch
