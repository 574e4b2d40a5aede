"""Single source of device random numbers for the training path (feature masks, shared
dropout masks, layer-skip masks).  Default: torch's generator of the target device.  Parity
tests swap `rand` for a CPU-generator version so that masks equal the oracle's."""
import torch


def rand(*shape, device=None, dtype=torch.float32):
    return torch.rand(*shape, device=device, dtype=dtype)
