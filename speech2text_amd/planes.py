"""bf16 pieces of a model's weight matrices, written once per optimizer step.

The forward (y = x W^T + b) and data-gradient (dx = g W) products of the layers' Linears
(reference model/encoder/zipformer.py:1924-1975, 2372-2378, 2643-2695) run on the bf16 matrix
cores with every fp32 operand split exactly into three bf16 pieces (csrc/gemm_x3p.hip).  The
weight side of that split does not depend on the batch: `PlaneArena` keeps, for every matrix
parameter of a `flat.FlatStore`, the pieces of W (for the forward) and of W^T (for the data
gradient) in the fragment-major order the kernel copies straight into LDS, and rewrites ALL of
them with one launch when the store's parameters have changed (`FlatStore.epoch`, bumped by the
fused optimizers and the DP broadcast; an in-place torch op on a parameter is caught through
its `_version`).  A weight that is not in a store has no entry and the caller takes the
library path.
"""
import ctypes
import weakref

import numpy as np
import torch

from . import _native as N
from . import flat

_DESC = np.dtype([("src_off", "<i8"), ("dst_off", "<i8"), ("N", "<i4"), ("K", "<i4"), ("ld", "<i4"),
                  ("transposed", "<i4"), ("blk_begin", "<u4"), ("pad", "<i4")])
assert _DESC.itemsize == 40

SPLITS = [0]         # refresh launches so far (tests)


class _Entry:
    __slots__ = ("param", "N", "K", "fwd", "bwd", "version")


class PlaneArena:
    def __init__(self, store):
        L = N.lib()
        self.store = weakref.proxy(store)
        self.entries = {}
        descs = []
        off = 0
        blk = 0
        for p, o in zip(store.params, store.offsets):
            if p.dim() < 2 or any(int(d) != 1 for d in p.shape[2:]):
                continue
            Nw, Kw = int(p.shape[0]), int(p.shape[1])
            if min(Nw, Kw) < 16:
                continue
            e = _Entry()
            e.param, e.N, e.K, e.fwd, e.bwd, e.version = p, Nw, Kw, None, None, p._version
            if Kw % 8 == 0 and Nw % 4 == 0:          # forward: Bm = W (N, K)
                e.fwd = off
                descs.append((o, off, Nw, Kw, Kw, 0, blk, 0))
                off += int(L.s2t_x3p_plane_elems(Nw, Kw))
                blk += int(L.s2t_x3p_split_blocks(Nw, Kw))
            if Nw % 8 == 0 and Kw % 4 == 0:          # data gradient: Bm = W^T (K, N)
                e.bwd = off
                descs.append((o, off, Kw, Nw, Kw, 1, blk, 0))
                off += int(L.s2t_x3p_plane_elems(Kw, Nw))
                blk += int(L.s2t_x3p_split_blocks(Kw, Nw))
            if e.fwd is not None or e.bwd is not None:
                self.entries[p.data_ptr()] = e
        self.ndesc, self.blocks = len(descs), blk
        dev = store.flat_p.device
        self.planes = torch.empty(max(off, 8), dtype=torch.int16, device=dev)
        tab = np.array(descs, dtype=_DESC) if descs else np.zeros(1, dtype=_DESC)
        self.tab = torch.from_numpy(tab.view(np.uint8).copy()).to(dev)
        self.epoch = -1

    def refresh(self):
        st = self.store
        if self.ndesc:
            N.PROF[0] and N.profile_note("s2t_x3p_split", 4.0 * st.numel + 2.0 * self.planes.numel())
            N.check(N.lib().s2t_x3p_split(N.fp(st.flat_p), ctypes.c_void_p(self.tab.data_ptr()),
                                          self.ndesc, self.blocks,
                                          ctypes.c_void_p(self.planes.data_ptr()), N.stream()),
                    "s2t_x3p_split")
            SPLITS[0] += 1
        for e in self.entries.values():
            e.version = e.param._version
        self.epoch = st.epoch

    def lookup(self, w2, mode):
        """-> device address of the pieces serving `mode` (0: x W^T, 1: g W) of the matrix view
        w2 of a parameter of this store, or None."""
        e = self.entries.get(w2.data_ptr())
        if e is None or e.N != w2.shape[0] or e.K != w2.shape[1] or w2.stride(0) != e.K:
            return None
        off = e.fwd if mode == 0 else e.bwd
        if off is None:
            return None
        if self.epoch != self.store.epoch or e.version != e.param._version:
            self.refresh()
        return self.planes.data_ptr() + 2 * off


def arena_of(w):
    """The arena of the store that owns the parameter whose storage `w` views, or None."""
    ref = flat._OWNER_PTR.get(w.data_ptr())
    store = None if ref is None else ref()
    if store is None:
        return None
    if store.arena is None:
        store.arena = PlaneArena(store)
    return store.arena


_HOT = {}            # (address, mode) -> (arena ref, entry, address of the pieces): the step's repeat lookups


def pieces(w2, mode):
    """Device address of the bf16 pieces of matrix w2 (a parameter's 2-D view) for `mode`, or None
    when the weight is not served (not in a FlatStore, shape outside the kernel's rules)."""
    key = (w2.data_ptr(), mode)
    hot = _HOT.get(key)
    if hot is not None:
        a, e, addr = hot[0](), hot[1], hot[2]
        if (a is not None and a.epoch == a.store.epoch and e.version == e.param._version
                and e.N == w2.shape[0] and e.K == w2.shape[1] and w2.stride(0) == e.K):
            return addr
        del _HOT[key]
    a = arena_of(w2)
    if a is None:
        return None
    addr = a.lookup(w2, mode)
    if addr is not None:
        if len(_HOT) > 4096:
            _HOT.clear()
        _HOT[key] = (weakref.ref(a), a.entries[w2.data_ptr()], addr)
    return addr


_ADHOC = {}          # (N, K, mode, device) -> (descriptor table, piece buffer, split workgroups)


def adhoc_entry(Nw, Kw, mode, device):
    """(descriptor table, piece buffer, split workgroups) of the ad-hoc piece slot of an (Nw, Kw)
    matrix for `mode`, created on first use; None when the shape is outside the kernel's rules."""
    if min(Nw, Kw) < 16:
        return None
    if not ((Kw % 8 == 0 and Nw % 4 == 0) if mode == 0 else (Nw % 8 == 0 and Kw % 4 == 0)):
        return None
    key = (Nw, Kw, mode, device.index)
    e = _ADHOC.get(key)
    if e is None:
        L = N.lib()
        n, k = (Nw, Kw) if mode == 0 else (Kw, Nw)
        tab = np.array([(0, 0, n, k, Kw, 1 if mode else 0, 0, 0)], dtype=_DESC)
        e = (torch.from_numpy(tab.view(np.uint8).copy()).to(device),
             torch.empty(max(int(L.s2t_x3p_plane_elems(n, k)), 8), dtype=torch.int16, device=device),
             int(L.s2t_x3p_split_blocks(n, k)))
        _ADHOC[key] = e
    return e


def adhoc_pieces(w2, mode):
    """Pieces of a matrix that is NOT a parameter (e.g. the Whiten penalty's d metric / d cov),
    written now, on the current stream, into a scratch buffer that the next call with the same
    shape reuses -- so the product that reads them must be enqueued on this stream before that
    call.  Returns the device address, or None when the shape is outside the kernel's rules."""
    if w2.dim() != 2 or w2.dtype is not torch.float32 or not w2.is_cuda or not w2.is_contiguous():
        return None
    Nw, Kw = int(w2.shape[0]), int(w2.shape[1])
    if w2.data_ptr() % 16:
        return None
    e = adhoc_entry(Nw, Kw, mode, w2.device)
    if e is None:
        return None
    tab, buf, blocks = e
    N.PROF[0] and N.profile_note("s2t_x3p_split", 4.0 * Nw * Kw + 2.0 * buf.numel())
    N.check(N.lib().s2t_x3p_split(w2.data_ptr(), tab.data_ptr(), 1, blocks, buf.data_ptr(), N.stream()),
            "s2t_x3p_split")
    return buf.data_ptr()

