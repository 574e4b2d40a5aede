"""Scratch experiment (GPU): where does the host time of a train step go?"""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from speech2text_amd.task_factory.rnnt_task import PrunedRnntTask
from speech2text_amd.trainer import Trainer
from speech2text_amd.model.layer import scaling

dev = torch.device("cuda", 0)
cfg = bench.c3_config(500)
torch.manual_seed(1234); random.seed(1234)
task = PrunedRnntTask(cfg)
trainer = Trainer(**cfg["trainer"]).setup(task, dev)
task.train()
batch = bench.make_batch(0, 64, 10.0, 50, 500, dev)

def run(tag, n=5):
    for i in range(2):
        trainer.training_step(batch, i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        trainer.training_step(batch, i)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print(f"{tag:40s} {1000*t/n:8.1f} ms/step   (host enqueue {1000*t_host/n:8.1f} ms)", flush=True)

run("baseline")
wf, bf = scaling.Whiten.fires, scaling.Balancer.fires
scaling.Whiten.fires = lambda self, x: False
run("whiten off")
scaling.Balancer.fires = lambda self, x: False
run("whiten+balancer off")
scaling.Whiten.fires, scaling.Balancer.fires = wf, bf
# forward only / forward+backward split
def fwd_only(n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        with torch.no_grad():
            task.training_step(batch, i)
    th = time.perf_counter() - t0
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f"{'forward only (no_grad)':40s} {1000*t/n:8.1f} ms/step   (host enqueue {1000*th/n:8.1f} ms)", flush=True)
fwd_only()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU]) as prof:
    trainer.training_step(batch, 0)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25))
