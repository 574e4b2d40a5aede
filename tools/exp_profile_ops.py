"""Scratch experiment (GPU): per-op GPU time of one train step (torch.profiler)."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from speech2text_amd.task_factory.rnnt_task import PrunedRnntTask
from speech2text_amd.trainer import Trainer
from torch.profiler import profile, ProfilerActivity, record_function

dev = torch.device("cuda", 0)
cfg = bench.c3_config(500)
torch.manual_seed(1234); random.seed(1234)
task = PrunedRnntTask(cfg)
trainer = Trainer(**cfg["trainer"]).setup(task, dev)
task.train()
batch = bench.make_batch(0, 64, 10.0, 50, 500, dev)
for i in range(3):
    trainer.training_step(batch, i)
torch.cuda.synchronize()

# module-level forward timing with events
ev = []
def mk(name):
    def pre(m, inp):
        e = torch.cuda.Event(enable_timing=True); e.record(); m._e0 = e
    def post(m, inp, out):
        e = torch.cuda.Event(enable_timing=True); e.record(); ev.append((name, m._e0, e))
    return pre, post
enc = task._encoder.encoder
mods = {"encoder_embed": enc._encoder_embed, "joiner": task._joiner, "predictor": task._predictor, "loss": task._loss}
for i, s in enumerate(enc.encoders):
    mods[f"stack{i}"] = s
hs = []
for n, m in mods.items():
    pre, post = mk(n)
    hs.append(m.register_forward_pre_hook(pre)); hs.append(m.register_forward_hook(post))
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True); t2 = torch.cuda.Event(enable_timing=True); t3 = torch.cuda.Event(enable_timing=True)
t0.record()
loss = task.training_step(batch, 0)
t1.record()
loss.backward()
t2.record()
trainer._clip(); trainer.optimizer.step(); trainer.scheduler.step(); trainer.store.zero_grad()
t3.record()
torch.cuda.synchronize()
print(f"forward {t0.elapsed_time(t1):.1f} ms  backward {t1.elapsed_time(t2):.1f} ms  clip+opt {t2.elapsed_time(t3):.1f} ms")
for n, a, b in ev:
    print(f"  fwd {n:16s} {a.elapsed_time(b):7.2f} ms")
for h in hs: h.remove()

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    trainer.training_step(batch, 1)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=60))
