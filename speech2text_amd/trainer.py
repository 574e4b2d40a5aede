"""Minimal trainer that drives a task the way pl.Trainer(**config["trainer"]).fit does in
the reference (build_task.py:143-148), honouring the YAML `trainer:` keys that matter for
the training hot path: devices/strategy (one process per GPU, RCCL), precision "32-true",
accumulate_grad_batches, gradient_clip_val / gradient_clip_algorithm, max_epochs.
"""
import os

import torch
import torch.distributed as dist

from speech2text_amd.ddp import GradReducer, broadcast_parameters
from speech2text_amd import zip_kernels as zk
from speech2text_amd.flat import get_store


class Trainer:
    def __init__(self, accelerator="gpu", devices=1, strategy="ddp", precision="32-true",
                 max_epochs=1, accumulate_grad_batches=1, gradient_clip_val=None,
                 gradient_clip_algorithm="norm", bucket_mb=32, **unused):
        if str(precision) not in ("32-true", "32"):
            raise NotImplementedError("the reference trains in fp32 ('32-true'); other "
                                      "precisions are not part of the parity target")
        self.accum = int(accumulate_grad_batches)
        self.clip_val = gradient_clip_val
        self.clip_algo = gradient_clip_algorithm
        self.max_epochs = max_epochs
        self.bucket_bytes = int(float(bucket_mb) * (1 << 20))
        self.micro = 0
        self.task = None
        self.log_every = int(unused.get("log_every_n_steps", 50))     # Lightning's default
        # Lightning's val_check_interval: a float = fraction of the training epoch between two
        # validation passes (1.0 = once at the end of every epoch), an int = every n training batches
        self.val_check_interval = unused.get("val_check_interval", 1.0)
        self.val_history = []          # one dict of averaged validation metrics per validation pass

    # ------------------------------------------------------------------
    def setup(self, task, device=None):
        if device is None:
            device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0))) \
                if torch.cuda.is_available() else torch.device("cpu")
        self.device = device
        if device.type == "cuda":
            torch.cuda.set_device(device)
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world > 1 and not dist.is_initialized():
            # one process per GPU (reference: pl.Trainer(strategy="ddp"), build_task.py:143-148);
            # backend "nccl" is RCCL on ROCm, gloo for the CPU rehearsals in tests/
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if device.type == "cuda":
                dist.init_process_group("nccl", device_id=device)
            else:
                dist.init_process_group("gloo")
        task.to(device)
        self.task = task
        opt = task.configure_optimizers()
        self.optimizer = opt["optimizer"]
        self.scheduler = opt["lr_scheduler"]["scheduler"]
        # ONE flat store per model, ordered by optimizer param group (so that each group is a
        # contiguous tensor range), then whatever trainable parameter no group lists
        seen, params = set(), []
        for group in self.optimizer.param_groups:
            for p in group["params"]:
                if p.requires_grad and id(p) not in seen:
                    seen.add(id(p))
                    params.append(p)
        for p in task.parameters():
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                params.append(p)
        self.store = get_store(params)
        self.fused = bool(getattr(self.optimizer, "fused_clip", False))
        if self.fused:
            if self.clip_val and self.clip_algo != "value":
                self.optimizer.pre_clip = float(self.clip_val)
            self.optimizer.zero_grad_in_step = True
        broadcast_parameters(self.store)
        self.reducer = GradReducer(self.store, self.bucket_bytes)
        return self

    def _clip(self):
        if not self.clip_val:
            return
        g = self.store.g()
        if self.clip_algo == "value":
            g.clamp_(-self.clip_val, self.clip_val)
            return
        if self.fused:
            return                                      # applied inside the optimizer's kernels
        norm = g.norm()
        g.mul_(torch.clamp(self.clip_val / (norm + 1.0e-6), max=1.0))

    def training_step(self, batch, batch_idx):
        """One micro-batch: forward, backward (+ overlapped gradient all-reduce on the last
        micro-batch of an accumulation window), then clip / optimizer / scheduler."""
        task = self.task
        last = (self.micro + 1) % self.accum == 0
        on_gpu = self.device.type == "cuda"
        if on_gpu:
            zk.side_sync()            # a backward that raised leaves the side stream un-joined
        if last:
            self.reducer.prepare()
            loss = task.training_step(batch, batch_idx)
            (loss / self.accum).backward()
        else:
            with self.reducer.no_sync():
                loss = task.training_step(batch, batch_idx)
                (loss / self.accum).backward()
        self.micro += 1
        if last:
            logged = None
            if self.reducer.active and task.logged:
                vals = [v.detach().float().reshape(()) if torch.is_tensor(v)
                        else torch.tensor(float(v), device=loss.device) for v in task.logged.values()]
                logged = torch.stack(vals)
            logged = self.reducer.finish(logged)
            if logged is not None:
                task.logged = dict(zip(task.logged.keys(), logged.unbind(0)))
            if on_gpu:
                zk.side_sync()        # weight gradients of this step are complete before we read
            self._clip()
            if self.reducer.active and hasattr(self.optimizer, "skip_flag"):
                self.optimizer.skip_flag = self.reducer.last_drop
            self.optimizer.step()
            self.scheduler.step()
            if not self.fused:
                self.store.zero_grad()
            task.global_step += 1
            if self.reducer.active and task.global_step % self.log_every == 0:
                # folds the recorded device flags (one host read per log_every steps)
                task.logged["dropped_steps"] = float(self.reducer.poll_dropped())
        return loss.detach()

    def validate(self, task, val_batches):
        """One validation pass (reference task_factory/*_task.py validation_step under Lightning's
        validation loop): eval mode, no gradients, every logged scalar averaged over the SAMPLES of
        the pass (Lightning's epoch-level log_dict reduction weights a batch by its size) -- and over
        the ranks when the job is data-parallel (sync_dist=True).  Every rank takes part in the
        reduction, also one whose shard of the validation set is empty.  Returns the averaged dict
        (empty when no rank saw a batch) and appends it to `val_history`."""
        was_training = task.training
        task.eval()
        sums, n = {}, 0.0
        with torch.no_grad():
            for i, batch in enumerate(val_batches):
                batch = {k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in batch.items()}
                out = task.validation_step(batch, i)
                vals = out if isinstance(out, dict) else dict(getattr(task, "logged", {}))
                bs = 1.0
                for v in batch.values():                       # batch size = leading dimension
                    if torch.is_tensor(v) and v.dim() >= 1:
                        bs = float(v.shape[0])
                        break
                for k, v in vals.items():
                    sums[k] = sums.get(k, 0.0) + float(v) * bs
                n += bs
        if was_training:
            task.train()
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if multi:
            # the metric names of a task are fixed (its `_metric` / loss keys): a rank without
            # batches learns them from the others so that all ranks reduce the same vector
            names = [None] * dist.get_world_size()
            dist.all_gather_object(names, sorted(sums))
            keys = sorted(set().union(*[set(x) for x in names]))
        else:
            keys = sorted(sums)
        t = torch.tensor([sums.get(k, 0.0) for k in keys] + [n], dtype=torch.float64)
        if multi:
            t = t.to(self.device) if dist.get_backend() == "nccl" else t
            dist.all_reduce(t)
            t = t.cpu()
        if float(t[-1]) == 0.0:
            return {}
        res = {k: float(t[i] / t[-1]) for i, k in enumerate(keys)}
        res["epoch"], res["global_step"] = task.current_epoch, task.global_step
        self.val_history.append(res)
        return res

    def _checkpoint(self, ck, res):
        """ModelCheckpoint(dirpath, filename, monitor, mode, save_top_k) after a validation pass
        (reference build_task.py:94-103): rank 0 writes `<name>-epoch=E-val_loss=L-<monitor>=S.ckpt`
        (the monitor field is ALWAYS appended, as the reference's filename template does -- also when
        the monitor is val_loss) when the monitored score enters the top-k table; the file that
        leaves the table is removed; a name already in the table gets Lightning's `-vN` suffix."""
        from speech2text_amd import checkpoint as C
        tracker = ck.get("_tracker")
        if tracker is None:
            cfg = ck.get("config") or {}
            tracker = ck["_tracker"] = C.BestK(monitor=cfg.get("monitor", "val_loss"),
                                               save_top_k=cfg.get("save_top_k", 3), mode=cfg.get("mode", "min"))
            tracker.best_k_models.update(ck.get("resumed_best_k") or {})
        if tracker.monitor not in res:                       # (raised on every rank alike)
            raise KeyError("checkpoint monitor %r is not among the validation metrics %s"
                           % (tracker.monitor, sorted(res)))
        if dist.is_available() and dist.is_initialized() and dist.get_rank() != 0:
            return None
        score = res[tracker.monitor]
        name = "%s-epoch=%d-val_loss=%.2f-%s=%.2f" % (ck.get("name", "task"), res["epoch"],
                                                      res.get("val_loss", float("nan")), tracker.monitor, score)
        path, v = os.path.join(ck["dirpath"], name + ".ckpt"), 0
        while path in tracker.best_k_models or os.path.exists(path):
            v += 1
            path = os.path.join(ck["dirpath"], "%s-v%d.ckpt" % (name, v))
        return path if C.save_checkpoint(self, path, score=score, tracker=tracker) else None

    def fit(self, task, batches, device=None, val_batches=None, checkpoint=None):
        """checkpoint (optional): {"dirpath": ..., "name": ..., "config": the YAML's
        callbacks.model_chkpt_config} -- a file is considered after every validation pass.
        After `checkpoint.resume` (an end-of-epoch file of epoch E) the loop continues at epoch
        E + 1 and stops at max_epochs, as `trainer.fit(ckpt_path=...)` does in the reference
        (build_task.py:148)."""
        if self.task is None:
            self.setup(task, device)
        task.train()
        vci = self.val_check_interval
        n_train = len(batches) if hasattr(batches, "__len__") else None
        if isinstance(vci, float) and n_train:
            every = max(1, int(round(vci * n_train)))          # fraction of an epoch
        elif isinstance(vci, int) and vci > 0:
            every = vci                                        # every n training batches
        else:
            every = None                                       # unknown length: at the end of the epoch
        for epoch in range(int(getattr(self, "start_epoch", 0)), self.max_epochs):
            task.current_epoch = epoch
            done = False
            for i, batch in enumerate(batches):
                batch = {k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in batch.items()}
                self.training_step(batch, i)
                done = False
                if val_batches is not None and every and (i + 1) % every == 0:
                    res = self.validate(task, val_batches)
                    if checkpoint is not None and res:
                        self._checkpoint(checkpoint, res)
                    done = True
            if val_batches is not None and not done:
                res = self.validate(task, val_batches)
                if checkpoint is not None and res:
                    self._checkpoint(checkpoint, res)
