"""joint_embedding_pretraining/tester.py of the reference: average loss over a dataloader in eval mode.
The reference's class reads `self.bfloat16` without ever setting it (tester.py:4-10 vs :47) although train.py:125
passes `bfloat16=`; the keyword is accepted here (decision recorded in DESIGN.md section 7)."""
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
        total_loss = 0
        num_lines = 0
        num_batches = 0
        self.model.eval()
        with torch.no_grad():
            for batch in self.dataloader:
                result = self.test_step(batch)
                total_loss = total_loss + result["loss"]
                num_lines += self.batch_operator.batch_size(batch)
                num_batches += 1
                if self.max_lines is not None and num_lines > self.max_lines:
                    break
        self.model.train()
        return {"loss": total_loss / num_batches}

    def test_step(self, batch):
        with autocast(self.bfloat16):
            return self.model.forward(*self.batch_operator.prepare_batch(batch))
