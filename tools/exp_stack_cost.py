"""Scratch experiment (GPU): marginal cost of one more layer in each zipformer stack at C3."""
import os, sys, random, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from speech2text_amd.task_factory.rnnt_task import PrunedRnntTask
from speech2text_amd.trainer import Trainer

dev = torch.device("cuda", 0)
batch = bench.make_batch(0, 64, 10.0, 50, 500, dev)


def run(layers):
    cfg = bench.c3_config(500)
    cfg["encoder"]["config"]["num_encoder_layers"] = list(layers)
    torch.manual_seed(1234); random.seed(1234)
    task = PrunedRnntTask(cfg)
    tr = Trainer(**cfg["trainer"]).setup(task, dev)
    task.train()
    for i in range(4):
        tr.training_step(batch, i)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(8):
            tr.training_step(batch, i)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 8 * 1e3)
    del tr, task
    torch.cuda.empty_cache()
    return best


base_layers = list(bench.c3_config(500)["encoder"]["config"]["num_encoder_layers"])
base = run(base_layers)
print(f"base {base_layers}: {base:.2f} ms/step", flush=True)
for s in range(len(base_layers)):
    l = list(base_layers); l[s] += 1
    t = run(l)
    print(f"stack {s} (+1 layer): {t:.2f} ms/step -> {t - base:.2f} ms per layer, x{base_layers[s]} = {(t - base) * base_layers[s]:.2f} ms", flush=True)
