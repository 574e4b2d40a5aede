"""BEST-RQ random-projection quantizer layer (mirror of the reference's model/ssl/best_rq.py).

Same configs, parameter names (`_projector`, `_codebooks.N`, frozen) and forward contract
`forward(raw_feats, auged_feats, length) -> {"masked_feats", "labels", "masked_dim"}`.
Labels come from the HIP kernel (fp64 inside, bit-exact integers); the span masks are drawn
on the host with numpy exactly as the reference does (same RNG calls in the same order, so a
fixed `seed` reproduces its masks); the noise fill runs on the GPU.
"""
import dataclasses
import math
from typing import Dict, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn as nn

from speech2text_amd import _native as N


@dataclasses.dataclass
class BestRQLayerConfig:
    pre_post_norm: bool = False
    cnn_kernel_size: Tuple = (3, 3)
    cnn_stride: Tuple = (2, 2)
    feat_dim: int = 80
    num_codebooks: int = 1
    codebook_dim: int = 512
    codebook_size: int = 1024
    label_basis: str = "euclidean"


@dataclasses.dataclass
class MaskingStrategyConfig:
    mask_proportion: float = 0.1
    mean_span_length: int = 1
    span_select_type: str = "static"
    span_length_float_rate: Union[int, float, None] = None
    min_num_spans: int = 0
    no_overlap: bool = False
    min_space: int = 0
    seed: Optional[int] = None


class BestRQLayer(nn.Module):
    def __init__(self, layer_config: BestRQLayerConfig, masking_config: MaskingStrategyConfig):
        super().__init__()
        if tuple(layer_config.cnn_kernel_size) != (3, 3) or tuple(layer_config.cnn_stride) != (2, 2):
            raise NotImplementedError("the label kernel is built for kernel (3,3) / stride (2,2) "
                                      "(the Subsampling(4) arrangement every shipped YAML uses)")
        if layer_config.pre_post_norm:
            raise NotImplementedError("pre_post_norm is deprecated upstream and unused by the YAMLs")
        assert layer_config.label_basis in ("euclidean", "cosine")
        self._label_basis = layer_config.label_basis
        self._num_codebooks = layer_config.num_codebooks
        self._codebook_dim = layer_config.codebook_dim
        self._codebook_size = layer_config.codebook_size
        self._codebooks = nn.ParameterList([
            nn.Parameter(torch.empty(self._codebook_size, self._codebook_dim), requires_grad=False)
            for _ in range(self._num_codebooks)])
        for cb in self._codebooks:
            nn.init.normal_(cb)
        self._feat_dim = layer_config.feat_dim
        self._input_dim = self._feat_dim * 9
        self._projector = nn.Parameter(torch.rand(self._input_dim, self._codebook_dim),
                                       requires_grad=False)
        nn.init.xavier_normal_(self._projector)
        m = masking_config
        self._mask_proportion, self._mean_span_length = m.mask_proportion, m.mean_span_length
        self._span_select_type, self._span_length_float_rate = m.span_select_type, m.span_length_float_rate
        self._min_num_spans, self._no_overlap, self._min_space = m.min_num_spans, m.no_overlap, m.min_space
        self._seed = m.seed

    @property
    def num_codebooks(self):
        return self._num_codebooks

    # ------------------------------------------------------------------ labels (GPU)
    @torch.no_grad()
    def make_labels(self, feats: torch.Tensor) -> torch.Tensor:
        if not feats.is_cuda:
            raise RuntimeError("BestRQLayer runs on the GPU only (no CPU fallback)")
        feats = feats.contiguous().float()
        B, T, F = feats.shape
        T2 = (((T - 3) // 2 + 1) - 3) // 2 + 1
        cbs = torch.stack([c.detach() for c in self._codebooks]).contiguous().float()
        labels = torch.empty((self._num_codebooks, B, T2), dtype=torch.int64, device=feats.device)
        N.check(N.lib().s2t_bestrq_labels(N.fp(feats), B, T, F,
                                          N.fp(self._projector.detach().contiguous().float()),
                                          self._codebook_dim, N.fp(cbs), self._num_codebooks,
                                          self._codebook_size, T2, N.lp(labels), N.stream()),
                "s2t_bestrq_labels")
        return labels

    # ------------------------------------------------------------------ masks (host, numpy)
    def _compute_mask_indices(self, timestep: int, padding_num: Optional[int]) -> np.ndarray:
        """Same algorithm and RNG call order as reference best_rq.py:296-405."""
        all_sz = timestep
        all_num_mask = int(self._mask_proportion * all_sz / float(self._mean_span_length)
                           + np.random.rand())
        all_num_mask = max(self._min_num_spans, all_num_mask)
        rng = np.random.default_rng(self._seed)
        if padding_num is not None:
            sz = all_sz - padding_num
            num_mask = int(self._mask_proportion * sz / float(self._mean_span_length) + rng.random())
            num_mask = max(self._min_num_spans, num_mask)
        else:
            sz, num_mask = all_sz, all_num_mask
        t = self._span_select_type
        if t == "static":
            lengths = np.full(num_mask, self._mean_span_length).tolist()
        elif t == "uniform":
            lengths = rng.integers(self._mean_span_length - self._span_length_float_rate,
                                   self._mean_span_length + self._span_length_float_rate,
                                   size=num_mask).tolist()
        elif t == "normal":
            lengths = [max(1, int(round(x))) for x in
                       rng.normal(self._mean_span_length, self._span_length_float_rate, size=num_mask)]
        elif t == "poisson":
            lengths = [int(round(x)) for x in rng.poisson(self._mean_span_length, size=num_mask)]
        else:
            raise Exception("unknown mask selection: " + t)
        if sum(lengths) == 0:
            lengths.append(min(self._mean_span_length, sz - 1))
        if self._no_overlap:
            idc = []

            def arrange(s, e, length, keep):
                start = s if s == e - length else rng.integers(s, e - length)
                idc.extend(start + i for i in range(length))
                parts = []
                if start - s - self._min_space >= keep:
                    parts.append((s, start - self._min_space + 1))
                if e - start - length - self._min_space > keep:
                    parts.append((start + length + self._min_space, e))
                return parts

            parts = [(0, sz)]
            min_length = min(lengths)
            for length in sorted(lengths, reverse=True):
                lens = np.fromiter((e - s if e - s >= length + self._min_space else 0
                                    for s, e in parts), np.int64)
                if np.sum(lens) == 0:
                    break
                c = rng.choice(len(parts), p=lens / np.sum(lens))
                s, e = parts.pop(c)
                parts.extend(arrange(s, e, length, min_length))
            idc = np.asarray(idc)
        else:
            min_len = min(lengths)
            if sz - min_len <= num_mask:
                min_len = sz - num_mask - 1
            starts = rng.choice(sz - min_len, num_mask, replace=False)
            idc = np.asarray([starts[j] + off for j in range(len(starts))
                              for off in range(lengths[j])])
        return np.unique(idc[idc < sz]) if len(idc) else np.zeros(0, np.int64)

    @torch.no_grad()
    def forward(self, raw_feats: torch.Tensor, auged_feats: torch.Tensor,
                length: torch.Tensor) -> Dict[str, torch.Tensor]:
        labels = self.make_labels(raw_feats)
        B, T, F = auged_feats.shape
        T2 = labels.shape[2]
        lab_len = length.detach().cpu().numpy()
        for _ in range(2):
            lab_len = (lab_len - 3) // 2 + 1
        masked_dim = np.zeros((B, T2), np.float32)
        frame_mask = np.zeros((B, T), bool)
        for b in range(B):
            idx = self._compute_mask_indices(T2, int(T2 - lab_len[b]))
            if idx.size:
                masked_dim[b, idx] = 1
                # label t2 covers frames 4*t2 .. 4*t2+6  (unique columns of the two unfolds)
                fr = (4 * idx[:, None] + np.arange(7)[None, :]).reshape(-1)
                frame_mask[b, np.unique(fr[fr < T])] = True
        fm = torch.from_numpy(frame_mask).to(auged_feats.device)
        noise = torch.normal(0.0, 0.1, size=auged_feats.shape, device=auged_feats.device,
                             dtype=auged_feats.dtype)
        auged_feats[fm] = noise[fm]                      # in place, as the reference
        return {"masked_feats": auged_feats, "labels": labels,
                "masked_dim": torch.from_numpy(masked_dim).to(auged_feats.device)}
