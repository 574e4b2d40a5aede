"""Flat parameter / gradient storage.

One contiguous fp32 buffer for the parameters of a model and one for their gradients
(`p.data` / `p.grad` become views; every tensor starts on a 16-byte boundary).  The optimizer
then runs three kernels over the flat buffers (csrc/optim.hip) instead of hundreds of per-tensor
launches, and the data-parallel reducer all-reduces slices of the same gradient buffer in place
-- no bucket copies.  Sized for one GPU's 288 GB: everything stays resident.

A model has ONE store (the trainer creates it, ordered by optimizer param group); optimizer
param groups are contiguous tensor ranges [seg_lo, seg_hi) of it.
"""
import weakref
from typing import List

import torch

_OWNER = {}          # id(param) -> (store, index)
_OWNER_PTR = {}      # device address of a parameter's storage -> weakref(store)  (planes.py)
CHUNK = 8192         # elements per optimizer workgroup (== s2t_optim_chunk_elems())
_PAD = 4             # tensors start on 16-byte boundaries (float4 lanes)


class FlatStore:
    def __init__(self, params: List[torch.nn.Parameter], align: int = 256):
        assert len(params) > 0
        dev, dt = params[0].device, params[0].dtype
        assert dt == torch.float32, "the training path is fp32 (precision: 32-true)"
        for p in params:
            if id(p) in _OWNER and _OWNER[id(p)][0].alive(p):
                raise RuntimeError("parameter already belongs to another FlatStore; a model has "
                                   "one store (param groups are ranges of it)")
        self.params = list(params)
        self.lengths = [p.numel() for p in self.params]
        self.offsets = []
        off = 0
        for n in self.lengths:
            self.offsets.append(off)
            off += ((n + _PAD - 1) // _PAD) * _PAD
        self.numel = off
        total = ((off + align - 1) // align) * align
        self.total = total
        self.flat_p = torch.zeros(total, dtype=dt, device=dev)
        self.flat_g = torch.zeros(total, dtype=dt, device=dev)
        with torch.no_grad():
            for i, (p, o, n) in enumerate(zip(self.params, self.offsets, self.lengths)):
                self.flat_p[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[o:o + n].view(p.shape)
                g = self.flat_g[o:o + n].view(p.shape)
                if p.grad is not None:
                    g.copy_(p.grad)
                p.grad = g
                _OWNER[id(p)] = (self, i)
                _OWNER_PTR[p.data_ptr()] = weakref.ref(self)
        self.seg_lengths = torch.tensor(self.lengths, dtype=torch.int64, device=dev)
        self._tables = None
        self.on_grad = None          # callback(index): set by the data-parallel reducer
        self.epoch = 0               # bumped by whoever rewrites flat_p through raw pointers
        self.arena = None            # planes.PlaneArena: bf16 pieces of the weight matrices

    def alive(self, p):
        i = _OWNER[id(p)][1]
        return i < len(self.params) and self.params[i] is p and \
            p.data_ptr() == self.flat_p.data_ptr() + 4 * self.offsets[i]

    # ---- chunk table for the fused optimizer kernels
    def tables(self):
        if self._tables is None:
            dev = self.flat_p.device
            c_off, c_len, c_seg, s_begin = [], [], [], [0]
            for s, (o, n) in enumerate(zip(self.offsets, self.lengths)):
                padded = ((n + _PAD - 1) // _PAD) * _PAD
                q = 0
                while q < padded:
                    ln = min(CHUNK, padded - q)
                    c_off.append(o + q)
                    c_len.append(ln)
                    c_seg.append(s)
                    q += ln
                s_begin.append(len(c_off))
            i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)   # noqa: E731
            self._tables = dict(chunk_off=i32(c_off), chunk_len=i32(c_len), chunk_seg=i32(c_seg),
                                seg_chunk_begin=i32(s_begin), seg_len=i32(self.lengths),
                                nchunks=len(c_off), nseg=len(self.lengths))
        return self._tables

    def range_of(self, params):
        """[lo, hi) tensor range of `params`, which must be a contiguous run of this store."""
        idx = []
        for p in params:
            st, i = _OWNER.get(id(p), (None, -1))
            if st is not self:
                raise RuntimeError("parameter is not in this FlatStore")
            idx.append(i)
        if idx != list(range(idx[0], idx[0] + len(idx))):
            raise RuntimeError("an optimizer param group must be a contiguous, ordered run of "
                               "the model's FlatStore (the trainer orders the store by group)")
        return idx[0], idx[0] + len(idx)

    def p(self):
        return self.flat_p[:self.numel]

    def g(self):
        return self.flat_g[:self.numel]

    def zero_grad(self):
        self.flat_g.zero_()

    def check_views(self):
        """Re-attach .grad views if something replaced them (e.g. zero_grad(set_to_none))."""
        for p, o, n in zip(self.params, self.offsets, self.lengths):
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * o:
                g = self.flat_g[o:o + n].view(p.shape)
                if p.grad is not None:
                    g.copy_(p.grad)
                p.grad = g


def owned(param) -> bool:
    ent = _OWNER.get(id(param))
    return ent is not None and ent[0].alive(param)


def grad_written(param):
    """A kernel accumulated this parameter's gradient directly into its flat view from OUTSIDE
    autograd's graph of that parameter (the layer executor, zip_layer.py: the parameter is not an
    input of its autograd node, so no AccumulateGrad / post-accumulate hook runs for it): tell the
    data-parallel reducer.  Autograd Functions that write a parameter's gradient in place and
    return None for it must NOT call this -- the engine still runs the parameter's
    post-accumulate hook afterwards, and that hook is the reducer's signal."""
    ent = _OWNER.get(id(param))
    if ent is not None and ent[0].on_grad is not None:
        ent[0].on_grad(ent[1])


def store_of(params):
    """The FlatStore that owns `params` (all of them), or None if none is owned yet."""
    params = list(params)
    owners = {id(_OWNER[id(p)][0]) for p in params if id(p) in _OWNER and _OWNER[id(p)][0].alive(p)}
    n_owned = sum(1 for p in params if id(p) in _OWNER and _OWNER[id(p)][0].alive(p))
    if n_owned == 0:
        return None
    if n_owned != len(params) or len(owners) != 1:
        raise RuntimeError("parameters are split across FlatStores / partly unowned")
    return _OWNER[id(params[0])][0]


def get_store(params) -> FlatStore:
    """The store owning `params`, created on first use."""
    params = list(params)
    st = store_of(params)
    return st if st is not None else FlatStore(params)
