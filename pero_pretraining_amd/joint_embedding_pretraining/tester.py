"""Evaluation loop of the joint-embedding model (surface of the reference's joint_embedding_pretraining/tester.py:
`Tester(batch_operator, model, dataloader, max_lines=None).test() -> {'loss': mean batch loss}`, `test_step(batch)`).

The reference's class reads `self.bfloat16` without ever setting it (tester.py:4-10 vs :47) although its train.py:125
passes `bfloat16=`; the keyword is accepted here (decision recorded in DESIGN.md section 7).  The loss stays a device tensor
through the loop: no per-batch synchronisation."""
import torch

from ..precision import autocast


class Tester:
    def __init__(self, batch_operator, model, dataloader, max_lines=None, bfloat16=False):
        self.batch_operator = batch_operator
        self.model = model
        self.dataloader = dataloader
        self.max_lines = max_lines
        self.bfloat16 = bfloat16

    def test(self):
        losses, lines_seen = [], 0
        self.model.eval()
        try:
            with torch.no_grad():
                for batch in self.dataloader:
                    losses.append(self.test_step(batch)["loss"])
                    lines_seen += self.batch_operator.batch_size(batch)
                    if self.max_lines is not None and lines_seen > self.max_lines:  # stops after the batch that exceeds it
                        break
        finally:
            self.model.train()
        return {"loss": sum(losses) / len(losses)}

    def test_step(self, batch):
        prepared = self.batch_operator.prepare_batch(batch)
        with autocast(self.bfloat16):
            return self.model.forward(*prepared)
