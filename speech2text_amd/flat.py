"""Flat parameter / gradient storage.

One contiguous fp32 buffer for the parameters of a group and one for their gradients
(`p.data` / `p.grad` become views).  The optimizer then runs a handful of kernels over the
flat buffers (segmented reductions for the per-tensor statistics) instead of hundreds of
per-tensor launches, and the data-parallel reducer all-reduces slices of the same gradient
buffer in place -- no bucket copies.  Sized for one GPU's 288 GB: everything stays resident.
"""
from typing import List

import torch

_STORES = {}


class FlatStore:
    def __init__(self, params: List[torch.nn.Parameter], align: int = 256, steal: bool = False):
        """steal=True: `p.grad` stays None between steps; autograd then just parks each freshly
        produced gradient tensor on the parameter (no accumulate kernel per parameter) and
        gather() adds them into the flat buffer with multi-tensor launches."""
        assert len(params) > 0
        self.steal = steal
        dev, dt = params[0].device, params[0].dtype
        self.params = list(params)
        self.lengths = [p.numel() for p in self.params]
        self.offsets = []
        off = 0
        for n in self.lengths:
            self.offsets.append(off)
            off += n
        self.numel = off
        total = ((off + align - 1) // align) * align
        self.total = total
        self.flat_p = torch.zeros(total, dtype=dt, device=dev)
        self.flat_g = torch.zeros(total, dtype=dt, device=dev)
        with torch.no_grad():
            for p, o, n in zip(self.params, self.offsets, self.lengths):
                self.flat_p[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[o:o + n].view(p.shape)
                g = self.flat_g[o:o + n].view(p.shape)
                if p.grad is not None:
                    g.copy_(p.grad)
                p.grad = None if steal else g
        self.seg_lengths = torch.tensor(self.lengths, dtype=torch.int64, device=dev)
        self._seg_ids = None

    @property
    def seg_ids(self):
        if self._seg_ids is None:
            n = len(self.lengths)
            self._seg_ids = torch.repeat_interleave(
                torch.arange(n, device=self.flat_p.device, dtype=torch.int32), self.seg_lengths)
        return self._seg_ids

    def p(self):
        return self.flat_p[:self.numel]

    def g(self):
        return self.flat_g[:self.numel]

    def seg_sum(self, x):
        return torch.segment_reduce(x, "sum", lengths=self.seg_lengths, unsafe=True)

    def zero_grad(self):
        self.flat_g.zero_()

    def gather(self, members=None):
        """steal mode: flat_g[slice] += p.grad for every (listed) parameter that holds a parked
        gradient, then drop it.  A few multi-tensor launches instead of one add per parameter."""
        if not self.steal:
            return
        idx = range(len(self.params)) if members is None else members
        dst, src = [], []
        for q in idx:
            p = self.params[q]
            if p.grad is not None:
                o = self.offsets[q]
                dst.append(self.flat_g[o:o + self.lengths[q]].view(p.shape))
                src.append(p.grad)
                p.grad = None
        if dst:
            torch._foreach_add_(dst, src)

    def check_views(self):
        """Re-attach .grad views if something replaced them (e.g. zero_grad(set_to_none))."""
        if self.steal:
            return
        for p, o, n in zip(self.params, self.offsets, self.lengths):
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * o:
                g = self.flat_g[o:o + n].view(p.shape)
                if p.grad is not None:
                    g.copy_(p.grad)
                p.grad = g


def get_store(params, steal: bool = False) -> FlatStore:
    params = [p for p in params]
    key = tuple(id(p) for p in params)
    st = _STORES.get(key)
    if st is None:
        st = FlatStore(params, steal=steal)
        _STORES[key] = st
    elif steal and not st.steal:
        st.steal = True
        for p in st.params:
            p.grad = None
    return st
