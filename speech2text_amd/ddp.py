"""Data-parallel gradient reduction over RCCL/xGMI (or gloo on CPU for tests).

Replaces the implicit torch-DDP that Lightning sets up for the reference
(build_task.py:143-148, YAML `strategy: ddp_find_unused_parameters_true`).  One process per
GPU; gradients live in one flat buffer (speech2text_amd.flat.FlatStore) cut into contiguous
buckets in reverse parameter order.  A bucket is all-reduced in place on a side HIP stream as
soon as autograd has produced its last gradient, overlapping the rest of backward; parameters
that never receive a gradient (e.g. Zipformer2EncoderLayer.bypass_scale, reference
zipformer.py:1011-1012) keep their pre-zeroed gradient and are learned as "static unused"
after the first step, which is what find_unused_parameters achieves by graph traversal.
xGMI is point-to-point (7 links per GPU): buckets are sized (default 32 MiB) so that each
collective is bandwidth- rather than latency-bound and few collectives are in flight.
"""
import contextlib
from typing import List, Optional

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, store, bucket_bytes: int = 32 << 20, process_group=None,
                 algo: str = "allreduce"):
        self.store = store
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.algo = algo
        self.require_sync = True
        self._is_cuda = store.flat_g.is_cuda
        self._stream = torch.cuda.Stream() if self._is_cuda else None
        # ---- buckets: contiguous slices of flat_g, walking parameters in reverse order
        n = len(store.params)
        cap = max(1, bucket_bytes // 4)
        self.buckets: List[List[int]] = []          # [start, end) element ranges
        self.param_bucket = [0] * n
        end = store.numel
        i = n - 1
        while i >= 0:
            start = end
            j = i
            while j >= 0 and (end - store.offsets[j]) <= cap:
                start = store.offsets[j]
                j -= 1
            if j == i:                               # single tensor larger than the cap
                start = store.offsets[i]
                j = i - 1
            b = len(self.buckets)
            for q in range(j + 1, i + 1):
                self.param_bucket[q] = b
            self.buckets.append([start, end])
            end = start
            i = j
        self.bucket_members = [[q for q in range(n) if self.param_bucket[q] == b]
                               for b in range(len(self.buckets))]
        self._expected = None                        # params that produce grads (learned)
        self._pending = None
        self._fired = set()
        self._launched = [False] * len(self.buckets)
        self._works = []
        self._hooks = []
        if self.world > 1:
            for q, p in enumerate(store.params):
                if p.requires_grad:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(q)))

    # ------------------------------------------------------------------
    def _make_hook(self, q):
        def hook(_p):
            if not self.require_sync:
                return
            self._fired.add(q)
            if self._pending is None:
                return
            b = self.param_bucket[q]
            if q in self._expected:
                self._pending[b] -= 1
                if self._pending[b] == 0 and not self._launched[b]:
                    self._launch(b)
        return hook

    def _launch(self, b):
        s, e = self.buckets[b]
        buf = self.store.flat_g[s:e]
        self._launched[b] = True
        self.store.gather(self.bucket_members[b])
        if self._is_cuda:
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                self._works.append((w, buf))
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg)
            buf.div_(self.world)

    def prepare(self):
        """Call before backward of a micro-step whose gradients must be synchronised."""
        self._fired = set()
        self._launched = [False] * len(self.buckets)
        self._works = []
        if self._expected is not None:
            self._pending = [sum(1 for q in m if q in self._expected)
                             for m in self.bucket_members]
        else:
            self._pending = None

    def finish(self, extra: Optional[torch.Tensor] = None):
        """After backward: reduce whatever was not launched by the hooks, wait, average.
        `extra` (1-D tensor of logged scalars) is mean-reduced along with the gradients."""
        if self.world == 1:
            self.store.gather()
            return extra
        for b in range(len(self.buckets)):
            if not self._launched[b]:
                self._launch(b)
        if self._is_cuda:
            main = torch.cuda.current_stream()
            with torch.cuda.stream(self._stream):
                for w, buf in self._works:
                    w.wait()
                    buf.div_(self.world)
                if extra is not None:
                    self._stream.wait_stream(main)          # `extra` was produced on the main stream
                    dist.all_reduce(extra, op=dist.ReduceOp.SUM, group=self.pg)
                    extra.div_(self.world)
            torch.cuda.current_stream().wait_stream(self._stream)
        elif extra is not None:
            dist.all_reduce(extra, op=dist.ReduceOp.SUM, group=self.pg)
            extra.div_(self.world)
        if self._expected is None or self._fired != self._expected:
            self._expected = set(self._fired)
        return extra

    @contextlib.contextmanager
    def no_sync(self):
        old = self.require_sync
        self.require_sync = False
        try:
            yield
        finally:
            self.require_sync = old


def broadcast_parameters(store, src: int = 0, process_group=None):
    if dist.is_initialized() and dist.get_world_size(process_group) > 1:
        dist.broadcast(store.flat_p, src=src, group=process_group)
