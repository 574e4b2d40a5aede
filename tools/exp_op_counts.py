"""Scratch experiment (GPU): aten op counts / host time / device time of one train step."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from speech2text_amd.task_factory.rnnt_task import PrunedRnntTask
from speech2text_amd.trainer import Trainer
from torch.profiler import profile, ProfilerActivity

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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    trainer.training_step(batch, 1)
    torch.cuda.synchronize()
ka = prof.key_averages()
print(ka.table(sort_by="self_cpu_time_total", row_limit=60, max_name_column_width=50))
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=70, max_name_column_width=40, max_shapes_column_width=60))
