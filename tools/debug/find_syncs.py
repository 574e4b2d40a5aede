"""Where does a C3 training step synchronise the host with the GPU?  torch's sync debug mode warns
at every synchronising call (pageable H2D copies, .item(), ...); the call sites are printed once."""
import os, sys, random, warnings, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from speech2text_amd.task_factory.rnnt_task import PrunedRnntTask
from speech2text_amd.trainer import Trainer

dev = torch.device("cuda", 0)
cfg = bench.c3_config(500)
torch.manual_seed(1234); random.seed(1234)
task = PrunedRnntTask(cfg)
trainer = Trainer(**cfg["trainer"]).setup(task, dev)
task.train()
batch = bench.make_batch(0, int(sys.argv[1]) if len(sys.argv) > 1 else 16, 10.0, 50, 500, dev)
for i in range(4):
    trainer.training_step(batch, i)
torch.cuda.synchronize()
sites = collections.Counter()


def hook(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if "speech2text_amd" in f.filename or "bench.py" in f.filename]
    key = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(st[-3:]))
    sites[(str(message)[:60], key)] += 1


warnings.showwarning = hook
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode(1)
trainer.training_step(batch, 5)
torch.cuda.set_sync_debug_mode(0)
torch.cuda.synchronize()
for (msg, key), n in sites.most_common():
    print(f"{n:3d} x {msg} | {key}")
