"""masked_pretraining/batch_operator.py of the reference, MI355X-native: the uint8 NHWC batch is moved to
the device as is (245 760 B per 40x2048 line instead of 983 040 B of float32) and the
`.float().permute(0,3,1,2) / 255` of the reference happens inside the HIP front-end kernel, fused with
masking and patch extraction.  `float_images=True` reproduces the reference's float NCHW tensor
(through the same kernels: the model accepts both)."""
import numpy as np
import torch


class BatchOperator:
    def __init__(self, device, masking_prob, float_images=False):
        self.device = device
        self.masking_prob = masking_prob
        self.float_images = float_images

    def prepare_batch(self, batch):
        return self._prepare_batch_images(batch), self._prepare_batch_labels(batch), self._create_mask(batch)

    def _prepare_batch_images(self, batch):
        images = torch.as_tensor(batch["images"]).to(self.device, non_blocking=True)  # numpy (reference) or device tensor (GPU BatchCreator)
        if self.float_images:  # reference layout; only used when a caller needs the float tensor itself
            images = images.float().permute(0, 3, 1, 2) / 255.0
        return images

    def _prepare_batch_labels(self, batch):
        return torch.as_tensor(batch["labels"]).to(self.device, non_blocking=True).long()

    def _create_mask(self, batch):
        # host numpy RNG, exactly as masked_pretraining/batch_operator.py:27-32 (returns a numpy int array)
        labels = batch["labels"]
        active_labels = (labels >= 0).astype(int)
        return (np.random.rand(*labels.shape) < self.masking_prob).astype(int) * active_labels

    @staticmethod
    def batch_size(batch):
        return batch["images"].shape[0]
