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
