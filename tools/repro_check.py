#!/usr/bin/env python3
"""Run-to-run reproducibility of the masked step: two runs from the same seeds, four steps each - losses and a parameter checksum."""
import sys, torch
sys.path.insert(0, "/root/repo")
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
res = []
for run in range(2):
    torch.manual_seed(0)
    model, opt, sched, trainer = bench.build(dev, True)
    batches = bench.synthetic(0, 256, dev)
    losses = []
    for i in range(4):
        sched.update_learning_rate(i + 1)
        torch.manual_seed(100 + i)
        b = batches[i % 2]
        losses.append(float(trainer.train_step_prepared(b["images"], b["labels_dev"], b["mask_dev"])))
    torch.cuda.synchronize()
    cs = float(sum(p.double().sum() for p in model.parameters()))
    res.append((losses, cs))
    del model, opt, trainer
print(res[0]); print(res[1])
print("loss diffs", [abs(a - b) for a, b in zip(res[0][0], res[1][0])], "checksum diff", abs(res[0][1] - res[1][1]))
