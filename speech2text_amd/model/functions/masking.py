"""Length -> mask helpers (reference model/functions/masking.py:158-184 and make_non_pad_mask)."""
import torch


def make_pad_mask(lengths: torch.Tensor, max_len: int = 0) -> torch.Tensor:
    """True at padded positions; (B, max_len)."""
    if max_len <= 0:
        max_len = int(lengths.max().item())
    ar = torch.arange(max_len, dtype=torch.int64, device=lengths.device)
    return ar.unsqueeze(0) >= lengths.unsqueeze(1)


def make_non_pad_mask(lengths: torch.Tensor) -> torch.Tensor:
    return ~make_pad_mask(lengths)
