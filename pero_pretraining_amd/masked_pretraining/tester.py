"""masked_pretraining/tester.py of the reference (Tester: average loss + top-k error on the masked positions),
MI355X-native: the reference copies the whole (B, S, V) logit tensor to the host for every batch and runs numpy
argmax / argsort row by row (tester.py:72-113); here `pero_label_rank` computes the label's rank where the logits
live and accumulates the error counters on the device with integer atomics - one small read-back per test()."""
import torch

from .. import ops
from ..precision import autocast


class Tester:
    def __init__(self, batch_operator, model, dataloader, max_lines=None, measured_errors=(1, 3, 10), bfloat16=False):
        if len(measured_errors) > ops.MAX_TOPK:
            raise ValueError(f"at most {ops.MAX_TOPK} measured errors")
        self.batch_operator = batch_operator
        self.model = model
        self.dataloader = dataloader
        self.max_lines = max_lines
        self.measured_errors = measured_errors
        self.bfloat16 = bfloat16
        self._ks = None
        self._counters = None

    def test(self):
        """tester.py:16-52: {'loss': mean of the batch losses, 'errors_k': wrong / masked positions}."""
        total_loss = 0
        num_lines = 0
        num_batches = 0
        self._counters = None
        self.model.eval()
        with torch.no_grad():
            for batch in self.dataloader:
                result = self.test_step(batch)
                total_loss = total_loss + result["loss"]
                self._update_errors(None, result, batch)
                num_lines += self.batch_operator.batch_size(batch)
                num_batches += 1
                if self.max_lines is not None and num_lines > self.max_lines:
                    break
        self.model.train()
        average_loss = total_loss / num_batches
        counts = self._counters.cpu().tolist()  # the only device -> host transfer of the loop
        errors = {f"errors_{k}": counts[1 + i] / counts[0] for i, k in enumerate(self.measured_errors)}
        return {"loss": average_loss, **errors}

    def test_step(self, batch):
        images, labels, mask = self.batch_operator.prepare_batch(batch)
        if not isinstance(mask, torch.Tensor):
            mask = torch.from_numpy(mask)
        mask = mask.to(images.device, non_blocking=True)
        with autocast(self.bfloat16):
            output = self.model.forward(images, labels, mask)
        batch["mask"] = mask
        batch["_device_labels"] = labels
        return output

    def _update_errors(self, errors, result, batch):
        """Accumulate the counters for one batch; `errors` (the reference's host dict) is unused, kept for the signature."""
        output = result["output"]
        dev = output.device
        if self._counters is None or self._counters.device != dev:
            self._counters = torch.zeros(1 + len(self.measured_errors), dtype=torch.int64, device=dev)
            self._ks = torch.tensor(list(self.measured_errors), dtype=torch.int32, device=dev)
        labels = batch.get("_device_labels")
        if labels is None:
            labels = torch.as_tensor(batch["labels"]).to(dev).long()
        mask = torch.as_tensor(batch["mask"]).to(dev).long()
        logits = output.reshape(-1, output.shape[-1])
        ops.label_rank(logits, labels.reshape(-1).contiguous(), mask.reshape(-1).contiguous(), self._ks, self._counters)
        return errors
