"""Batch operator of the masked pre-training step (surface of the reference's masked_pretraining/batch_operator.py:
`prepare_batch(batch) -> (images, labels, mask)`, `batch_size(batch)`, `.device`, `.masking_prob`).

MI355X-first data flow: the uint8 NHWC batch goes to the device AS IS - 245 760 B per 40x2048 line instead of the
983 040 B of the float NCHW tensor the reference builds - and the `.float().permute(0, 3, 1, 2) / 255` happens inside the
HIP front-end kernel, fused with masking and patch extraction.  `float_images=True` returns the reference's float tensor
for callers that need it (the model accepts both).  Inputs may be numpy arrays (reference dataloader) or device tensors
(common/dataloader.BatchCreator).  The masking pattern is drawn on the HOST from numpy's global stream, in the reference's
order (batch_operator.py:27-32), so a seeded run masks the same positions."""
import numpy as np
import torch


def _to_device(value, device):
    return torch.as_tensor(value).to(device, non_blocking=True)


class BatchOperator:
    def __init__(self, device, masking_prob, float_images=False):
        self.device = device
        self.masking_prob = masking_prob
        self.float_images = float_images

    def prepare_batch(self, batch):
        return (self._prepare_batch_images(batch), self._prepare_batch_labels(batch), self._create_mask(batch))

    def _prepare_batch_images(self, batch):
        pixels = _to_device(batch["images"], self.device)
        if not self.float_images:
            return pixels
        return pixels.float().permute(0, 3, 1, 2) / 255.0

    def _prepare_batch_labels(self, batch):
        return _to_device(batch["labels"], self.device).long()

    def _create_mask(self, batch):
        """(N, S) int array: 1 where a position is masked; never on padding positions (label < 0)."""
        labels = np.asarray(batch["labels"])
        drawn = np.random.rand(*labels.shape) < self.masking_prob
        return drawn.astype(int) * (labels >= 0).astype(int)

    @staticmethod
    def batch_size(batch):
        return len(batch["images"])
