import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import bench
from pero_pretraining_amd.masked_pretraining.trainer import Trainer
dev = torch.device("cuda", 0)
model, opt, sched, trainer = bench.build(dev, True)
batches = bench.synthetic(0, 16, dev)
ref = Trainer(None, model, None, opt, sched, bfloat16=True)
sched.update_learning_rate(1)
l0 = float(ref.train_step_prepared(*batches[0]))
g = Trainer(None, model, None, opt, sched, bfloat16=True, hip_graph=True)
l1 = float(g.train_step_prepared(*batches[1])); l2 = float(g.train_step_prepared(*batches[0])); l3 = float(g.train_step_prepared(*batches[1]))
print("eager", l0, "graph", l1, l2, l3)
assert all(np.isfinite(v) for v in (l0, l1, l2, l3))
print("graph ok")
