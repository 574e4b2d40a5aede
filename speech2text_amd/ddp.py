"""Data-parallel gradient reduction over RCCL/xGMI (or gloo on CPU for tests).

Replaces the implicit torch-DDP that Lightning sets up for the reference
(build_task.py:143-148, YAML `strategy: ddp_find_unused_parameters_true`).  One process per
GPU; gradients live in one flat buffer (speech2text_amd.flat.FlatStore) cut into contiguous
buckets in reverse parameter order.  A bucket is reduced in place on a side HIP stream as soon
as autograd has produced its last gradient, overlapping the rest of backward; buckets are
launched strictly in bucket order so every rank issues the same sequence of collectives.
Parameters that never receive a gradient (e.g. Zipformer2EncoderLayer.bypass_scale, reference
zipformer.py:1011-1012) keep their pre-zeroed gradient and are learned as "static unused"
after the first step, which is what find_unused_parameters achieves by graph traversal.
The learned set only GROWS (per-parameter maximum of the calls seen in any step): a parameter
that skips a step defers its bucket (and the later ones) to finish(); a parameter that fires
for the first time AFTER its bucket went out cannot be merged into a collective that is already
in flight, so that step's gradient is zeroed on EVERY rank (the ranks agree through a flag that
rides in the step's scalar all-reduce -- no rank raises, none is left waiting in a collective),
the same device flag makes the optimizer step a no-op on parameters and state (`last_drop` ->
`optimizer.skip_flag`, read by the kernels, no host sync; the host-side step count and the LR
schedule still advance by one), `dropped_steps` counts it (the trainer logs it), and from the
next step on the parameter is waited for.

Two forms of the exchange (`algo`):
  "allreduce"  one in-place all-reduce(SUM) per bucket, then a divide over the whole bucket;
  "rs_ag"      reduce-scatter into a 1/world shard, divide the shard, all-gather back -- the form
               SURVEY.md section 5 prices for the 7-link xGMI mesh (each GPU owns 1/8 of a bucket).
xGMI is point-to-point (7 links per GPU): buckets are sized (default 32 MiB) so that each
collective is bandwidth- rather than latency-bound and few collectives are in flight.
"""
import contextlib
import os
import time
from typing import List, Optional

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, store, bucket_bytes: int = 32 << 20, process_group=None,
                 algo: Optional[str] = None, overlap: Optional[bool] = None,
                 force_active: Optional[bool] = None):
        self.store = store
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # force_active: run the whole machinery (hooks, buckets, side stream, collectives) on a
        # 1-rank group too -- a numerical no-op used to exercise RCCL on a single GPU
        if force_active is None:
            force_active = os.environ.get("S2T_DDP_FORCE", "0") == "1"
        self.active = self.world > 1 or (bool(force_active) and dist.is_initialized())
        self.algo = algo or os.environ.get("S2T_DDP_ALGO", "allreduce")
        if self.algo not in ("allreduce", "rs_ag"):
            raise ValueError(f"unknown gradient-exchange algo {self.algo!r}")
        self.overlap = (os.environ.get("S2T_DDP_OVERLAP", "1") != "0") if overlap is None \
            else bool(overlap)
        self.require_sync = True
        self._is_cuda = store.flat_g.is_cuda
        self._stream = torch.cuda.Stream() if self._is_cuda else None
        # ---- buckets: contiguous slices of flat_g, walking parameters in reverse order
        n = len(store.params)
        cap = max(1, bucket_bytes // 4)
        self.buckets: List[List[int]] = []          # [start, end) element ranges
        self.param_bucket = [0] * n
        end = store.total                            # the tail padding rides in bucket 0
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
        self._shards = {}
        self._expected = None                        # param -> max #calls per step (learned, monotone)
        self._pending = None
        self._fired = {}
        self._late = set()                           # fired after their bucket was launched
        self.dropped_steps = 0                       # steps whose gradient was zeroed (see above)
        self._drop_flags = []                        # device flags of recent steps (read lazily)
        self.last_drop = None                        # device flag of the last finish(): > 0 = dropped
        self._next = 0                               # next bucket to launch (fixed order)
        # overlap evidence of the last synchronised step: (bucket, host time, launched by a gradient
        # signal inside backward?) per collective, and the host time finish() was entered
        self.launch_log = []
        self.finish_t = None
        self._in_finish = False
        self._hooks = []
        if self.active:
            store.on_grad = self._grad_ready
            for q, p in enumerate(store.params):
                if p.requires_grad:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(q)))

    # ------------------------------------------------------------------
    def _make_hook(self, q):
        return lambda _p: self._grad_ready(q)

    def _grad_ready(self, q):
        """Parameter q's gradient of this backward is complete: called by autograd's
        post-accumulate hook (once per backward; it also runs when a Function wrote the gradient
        in place and returned None), or by the layer executor, whose parameters are not part of
        the autograd graph (FlatStore.on_grad, once per use).  The number of calls per parameter
        and step is learned in the first synchronised step."""
        if not self.require_sync:
            return
        self._fired[q] = self._fired.get(q, 0) + 1
        if self._pending is None:
            return
        b = self.param_bucket[q]
        if b < self._next:
            # the bucket is already with the exchange stream: this contribution cannot join it.
            # Never raise on one rank (the others would hang in the collective): finish() makes
            # all ranks agree to zero this step's gradient; the parameter joins the expected set.
            self._late.add(q)
            return
        if self._fired[q] <= self._expected.get(q, 0):
            self._pending[b] -= 1
            while self._next < len(self.buckets) and self._pending[self._next] == 0:
                self._launch(self._next)

    def _shard(self, b, n):
        sh = self._shards.get(b)
        if sh is None or sh.numel() != n:
            sh = torch.empty(n, dtype=self.store.flat_g.dtype, device=self.store.flat_g.device)
            self._shards[b] = sh
        return sh

    def _exchange(self, buf, b):
        """In-place mean over ranks of `buf` (enqueued on the side stream when on the GPU)."""
        w = self.world
        if self.algo == "rs_ag" and buf.numel() % w == 0 and buf.numel() >= w:
            sh = self._shard(b, buf.numel() // w)
            dist.reduce_scatter_tensor(sh, buf, op=dist.ReduceOp.SUM, group=self.pg)
            sh.div_(w)
            dist.all_gather_into_tensor(buf, sh, group=self.pg)
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg)
            buf.div_(w)

    def _launch(self, b):
        assert b == self._next
        self.launch_log.append((b, time.perf_counter(), not self._in_finish))
        s, e = self.buckets[b]
        buf = self.store.flat_g[s:e]
        self._next = b + 1
        if self._is_cuda:
            self._stream.wait_stream(torch.cuda.current_stream())
            self._wait_wgrad_stream()
            with torch.cuda.stream(self._stream):
                self._exchange(buf, b)
        else:
            self._exchange(buf, b)

    def _wait_wgrad_stream(self):
        """Weight gradients may still be in flight on the kernels' side stream (zip_kernels._Side):
        the exchange stream must be ordered after it as well."""
        from speech2text_amd import _native as N
        from speech2text_amd import zip_kernels as zk
        h = zk.side_stream_handle()
        if h is not None:
            import ctypes
            N.check(N.lib().s2t_stream_order(h, ctypes.c_void_p(self._stream.cuda_stream)),
                    "s2t_stream_order(reducer)")

    def prepare(self):
        """Call before backward of a micro-step whose gradients must be synchronised."""
        self._fired = {}
        self._late = set()
        self._next = 0
        self.launch_log = []
        self.finish_t = None
        if self._expected is not None and self.overlap:
            self._pending = [sum(self._expected.get(q, 0) for q in m)
                             for m in self.bucket_members]
        else:
            self._pending = None

    def finish(self, extra: Optional[torch.Tensor] = None):
        """After backward: reduce whatever was not launched by the hooks, wait, average.
        `extra` (1-D tensor of logged scalars) is mean-reduced along with the gradients."""
        self.last_drop = None
        if not self.active:
            return extra
        self.finish_t = time.perf_counter()
        self._in_finish = True
        try:
            while self._next < len(self.buckets):
                self._launch(self._next)
        finally:
            self._in_finish = False
        # one small all-reduce per step: [logged scalars ..., "some rank fired late" flag]
        fg = self.store.flat_g
        flag = torch.full((1,), 1.0 if self._late else 0.0, dtype=fg.dtype, device=fg.device)
        n_extra = 0 if extra is None else extra.numel()
        pack = flag if extra is None else torch.cat([extra.reshape(-1).to(fg.dtype), flag])
        if self._is_cuda:
            main = torch.cuda.current_stream()
            with torch.cuda.stream(self._stream):
                self._stream.wait_stream(main)          # `pack` was produced on the main stream
                dist.all_reduce(pack, op=dist.ReduceOp.SUM, group=self.pg)
            main.wait_stream(self._stream)
        else:
            dist.all_reduce(pack, op=dist.ReduceOp.SUM, group=self.pg)
        if self.overlap and self._pending is not None:
            # zero the step on every rank if any rank fired late (device-side: no host sync)
            # (masked fill, not a multiply: the late write raced with the collective in flight, so
            # the buffer can hold inf / NaN, and 0 * NaN stays NaN)
            self.last_drop = pack[n_extra:]
            fg.masked_fill_(self.last_drop > 0, 0.0)
            self._drop_flags.append(self.last_drop)
            if len(self._drop_flags) > 64:
                self.poll_dropped()
        if extra is not None:
            extra = (pack[:n_extra] / self.world).reshape(extra.shape).to(extra.dtype)
        # the expected set only grows: per-parameter maximum of the calls seen in a step
        if self._expected is None:
            self._expected = dict(self._fired)
        else:
            for q, c in self._fired.items():
                if c > self._expected.get(q, 0):
                    self._expected[q] = c
        return extra

    def overlap_report(self):
        """Of the last synchronised step: how many of the buckets went to the exchange stream from
        inside backward (a gradient signal completed them) rather than from finish(), and how long
        before finish() the first one did (host clock, ms)."""
        early = [t for (_, t, in_bwd) in self.launch_log if in_bwd]
        lead = None if (not early or self.finish_t is None) else 1e3 * (self.finish_t - early[0])
        return {"buckets": len(self.buckets), "launched_in_backward": len(early),
                "first_launch_ms_before_finish": lead}

    def poll_dropped(self):
        """Folds the recorded device flags into `dropped_steps` (a host read: call it off the
        hot path, e.g. when logging)."""
        if self._drop_flags:
            self.dropped_steps += int((torch.cat(self._drop_flags) > 0).sum().item())
            self._drop_flags = []
        return self.dropped_steps

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
        store.epoch += 1                   # parameters rewritten: the weights' bf16 pieces are stale
