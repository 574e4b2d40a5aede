import os, sys
sys.path.insert(0, os.getcwd())
import bench, torch, random
from speech2text_amd import _native as N
from speech2text_amd.task_factory.rnnt_task import PrunedRnntTask
from speech2text_amd.trainer import Trainer
dev = torch.device("cuda", 0)
cfg = bench.c3_config(500)
torch.manual_seed(1234); random.seed(1234)
task = PrunedRnntTask(cfg); trainer = Trainer(**cfg["trainer"]).setup(task, dev); task.train()
batch = bench.make_batch(0, 64, 10.0, 50, 500, dev)
for i in range(3): trainer.training_step(batch, i)
torch.cuda.synchronize()
c0 = N.lib().s2t_linear_lt_own_calls()
trainer.training_step(batch, 3); torch.cuda.synchronize()
print("own-kernel launches per step:", N.lib().s2t_linear_lt_own_calls() - c0, "of 432")
