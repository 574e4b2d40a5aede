"""GPU: per-step wall time of the C3 step with the YAML's random chunk sizes (synchronised after
every step): where do the occasional slow steps of `bench.py --random-chunk` come from?"""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from speech2text_amd.task_factory.rnnt_task import PrunedRnntTask
from speech2text_amd.trainer import Trainer
from speech2text_amd import zip_kernels as zk, zip_layer

dev = torch.device("cuda", 0)
cfg = bench.c3_config(500)
cfg["encoder"]["config"].update({"chunk_size": [16, 32, 64, -1], "left_context_frames": [64, 128, 256, -1]})
torch.manual_seed(1234); random.seed(1234)
task = PrunedRnntTask(cfg)
trainer = Trainer(**cfg["trainer"]).setup(task, dev)
task.train()
batch = bench.make_batch(0, 64, 10.0, 50, 500, dev)
for i in range(40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    p0, pa0 = zk.PLAN_STATS["timed"], zip_layer.STATS["penalty_active"]
    c0 = zip_layer.CALLS[0]
    trainer.training_step(batch, i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    print(f"step {i:2d}: {dt:7.1f} ms  plans timed +{zk.PLAN_STATS['timed'] - p0}  penalty_active +{zip_layer.STATS['penalty_active'] - pa0}"
          f"  executor calls +{zip_layer.CALLS[0] - c0}  mem {torch.cuda.memory_reserved() >> 20} MiB", flush=True)
