"""Checkpoint files in the layout the reference's Lightning run writes, so that files interchange
in both directions:

* `save_checkpoint` / `resume`: what `ModelCheckpoint` + `trainer.fit(ckpt_path=config["resume"])`
  do in the reference (build_task.py:97-103,148): `state_dict` with the task's attribute prefixes
  (`_encoder.…`, `_predictor.…`, `_joiner.…`), `optimizer_states`, `lr_schedulers`, `epoch`,
  `global_step` and the checkpoint callback's `best_k_models` table.
* `load_from_checkpoint`: the finetune path (build_task.py:82-92, `strict=False`).
* `model_average`: top-k averaging, reference tools/model_average.py:36-66 -- same inputs
  (a directory of `*.ckpt`), same selection (the newest file's `best_k_models` table, sorted by
  score), same output file name (`averaged.chkpt`) and arithmetic (sum in the stored dtype in pool
  order, then true division).

Plain `torch.save` / `torch.load` on CPU tensors: nothing here touches the GPU.
"""
import glob
import os

import torch

CALLBACK_KEY = "ModelCheckpoint"          # Lightning appends the callback's arguments to this


def _cpu(obj):
    if torch.is_tensor(obj):
        return obj.detach().cpu().clone()
    if isinstance(obj, dict):
        return {k: _cpu(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_cpu(v) for v in obj)
    return obj


class BestK:
    """The bookkeeping of `ModelCheckpoint(monitor=…, save_top_k=k, mode=…)`: path -> score of
    the k best files so far; files that fall out of the table are deleted."""

    def __init__(self, monitor="val_loss", save_top_k=3, mode="min"):
        if mode not in ("min", "max"):
            raise ValueError("mode is 'min' or 'max'")
        self.monitor, self.k, self.mode = monitor, int(save_top_k), mode
        self.best_k_models = {}

    def key(self):
        return "%s{'monitor': %r, 'mode': %r, 'save_top_k': %d}" % (CALLBACK_KEY, self.monitor,
                                                                     self.mode, self.k)

    def _worst(self):
        pick = max if self.mode == "min" else min
        return pick(self.best_k_models, key=lambda p: float(self.best_k_models[p]))

    def wants(self, score):
        if self.k < 0 or len(self.best_k_models) < self.k:
            return self.k != 0
        w = float(self.best_k_models[self._worst()])
        return score < w if self.mode == "min" else score > w

    def add(self, path, score):
        """Registers `path`; returns the path pushed out of the table (to delete) or None."""
        self.best_k_models[path] = torch.tensor(float(score))
        if self.k >= 0 and len(self.best_k_models) > self.k:
            out = self._worst()
            del self.best_k_models[out]
            return out
        return None

    def state(self):
        best = None
        if self.best_k_models:
            pick = min if self.mode == "min" else max
            best = pick(self.best_k_models, key=lambda p: float(self.best_k_models[p]))
        return {"monitor": self.monitor, "best_k_models": dict(self.best_k_models),
                "best_model_path": best or "",
                "best_model_score": self.best_k_models.get(best) if best else None}


def save_checkpoint(trainer, path, score=None, tracker=None):
    """Writes the trainer's task / optimizer / scheduler to `path`.  With a `BestK` tracker and a
    monitored `score` the file enters the top-k table (and the file that leaves it is removed);
    returns True when the file was written."""
    if tracker is not None and score is not None and not tracker.wants(float(score)):
        return False
    task = trainer.task
    if tracker is not None and score is not None:
        dropped = tracker.add(path, score)
        if dropped and dropped != path and os.path.exists(dropped):
            os.remove(dropped)
    ck = {"epoch": int(task.current_epoch), "global_step": int(task.global_step),
          "state_dict": _cpu(task.state_dict()),
          "optimizer_states": [_cpu(trainer.optimizer.state_dict())],
          "lr_schedulers": [_cpu(trainer.scheduler.state_dict())],
          "callbacks": {tracker.key(): tracker.state()} if tracker is not None else {}}
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(ck, path)
    return True


def load_from_checkpoint(task, path, strict=False):
    """Finetune start (reference build_task.py:82-92): parameters by name, nothing else.
    Returns torch's (missing_keys, unexpected_keys)."""
    ck = torch.load(path, map_location="cpu", weights_only=False)
    return task.load_state_dict(ck["state_dict"], strict=strict)


def resume(trainer, path):
    """`trainer.fit(ckpt_path=…)` (reference build_task.py:148): parameters, optimizer moments,
    scheduler position, epoch / step counters.  The trainer must be set up (flat store built)."""
    if trainer.task is None:
        raise RuntimeError("call Trainer.setup(task) before resume()")
    ck = torch.load(path, map_location="cpu", weights_only=False)
    trainer.task.load_state_dict(ck["state_dict"], strict=True)
    if ck.get("optimizer_states"):
        trainer.optimizer.load_state_dict(ck["optimizer_states"][0])
    if ck.get("lr_schedulers"):
        trainer.scheduler.load_state_dict(ck["lr_schedulers"][0])
    trainer.task.current_epoch = int(ck.get("epoch", 0))
    trainer.task.global_step = int(ck.get("global_step", 0))
    # the files ModelCheckpoint writes are end-of-validation files of epoch E: fit continues at
    # E + 1 and stops at max_epochs (Lightning's fit loop after ckpt_path)
    trainer.start_epoch = trainer.task.current_epoch + 1
    return ck


def resumed_best_k(ck):
    """The checkpoint callback's path -> score table stored in a resumed file (all callbacks'
    entries merged), for the new run's BestK tracker: files written before the resume stay in the
    top-k bookkeeping (eviction, `model_average`'s pool)."""
    table = {}
    for state in (ck.get("callbacks") or {}).values():
        table.update(state.get("best_k_models") or {})
    return table


def _latest(chkpt_dir):
    files = glob.glob(os.path.join(chkpt_dir, "*.ckpt"))
    if not files:
        raise FileNotFoundError("no *.ckpt under %s" % chkpt_dir)
    return max(files, key=os.path.getctime)


def _pool(latest, num_aver, descending):
    pool = []
    for state in latest["callbacks"].values():
        for path, score in state.get("best_k_models", {}).items():
            pool.append((path, float(score)))
    pool.sort(key=lambda e: e[1], reverse=descending)          # stable, like the reference's sort
    return pool if num_aver is None else pool[:num_aver]


def model_average(chkpt_dir, aver_best_k=None, descending=False):
    """Averages the `state_dict`s of the best checkpoints into `<chkpt_dir>/averaged.chkpt`
    (kept if it already exists).  `descending=True` when the monitored value is an accuracy.
    Returns the output path."""
    if not os.path.isdir(chkpt_dir):
        raise FileNotFoundError(chkpt_dir)
    out_path = os.path.join(chkpt_dir, "averaged.chkpt")
    if os.path.exists(out_path):
        return out_path
    latest = torch.load(_latest(chkpt_dir), map_location="cpu", weights_only=False)
    pool = _pool(latest, aver_best_k, descending)
    if not pool:
        raise ValueError("the newest checkpoint lists no best_k_models")
    total = None
    for path, _ in pool:
        ck = torch.load(path, map_location="cpu", weights_only=False)
        if total is None:
            total = ck
        else:
            for k, v in ck["state_dict"].items():
                total["state_dict"][k] += v
    for k in total["state_dict"]:
        total["state_dict"][k] = torch.true_divide(total["state_dict"][k], len(pool))
    torch.save(total, out_path)
    return out_path
