"""joint_embedding_pretraining/batch_operator.py of the reference: two image tensors + four u8 mask tensors to
the device.  Images stay uint8 NHWC (the /255 + NCHW permute is fused into the HIP front end) unless
float_images=True."""
import torch


class BatchOperator:
    def __init__(self, device, float_images=False):
        self.device = device
        self.float_images = float_images

    def prepare_batch(self, batch):
        return (self._prepare_batch_images(batch, "images"), self._prepare_batch_images(batch, "images2"),
                self._prepare_batch_masks(batch, "image_masks"), self._prepare_batch_masks(batch, "image_masks2"),
                self._prepare_batch_masks(batch, "shift_masks"), self._prepare_batch_masks(batch, "shift_masks2"))

    def _prepare_batch_images(self, batch, key="images"):
        images = torch.as_tensor(batch[key]).to(self.device, non_blocking=True)  # numpy (reference) or device tensor (GPU BatchCreator)
        if self.float_images:
            images = images.float().permute(0, 3, 1, 2) / 255.0
        return images

    def _prepare_batch_masks(self, batch, key="image_masks"):
        return torch.as_tensor(batch[key]).to(self.device, non_blocking=True)

    @staticmethod
    def batch_size(batch):
        return batch["images"].shape[0]
