"""Batch operator of the joint-embedding step (surface of the reference's joint_embedding_pretraining/batch_operator.py):
`prepare_batch(batch)` -> (images1, images2, image_masks1, image_masks2, shift_masks1, shift_masks2) on the device.
Images stay uint8 NHWC (the /255 and the NCHW permute are fused into the HIP front end) unless `float_images=True`;
masks stay uint8.  Values may be numpy arrays or device tensors (common/dataloader.BatchCreator)."""
import numpy as np
import torch

_IMAGE_KEYS = ("images", "images2")
_MASK_KEYS = ("image_masks", "image_masks2", "shift_masks", "shift_masks2")


class BatchOperator:
    def __init__(self, device, float_images=False):
        self.device = device
        self.float_images = float_images

    def prepare_batch(self, batch):
        views = [self._prepare_batch_images(batch, key) for key in _IMAGE_KEYS]
        masks = [self._prepare_batch_masks(batch, key) for key in _MASK_KEYS]
        return (*views, *masks)

    def _prepare_batch_images(self, batch, key="images"):
        pixels = torch.as_tensor(batch[key]).to(self.device, non_blocking=True)
        return pixels.float().permute(0, 3, 1, 2) / 255.0 if self.float_images else pixels

    def _prepare_batch_masks(self, batch, key="image_masks"):
        value = batch[key]
        t = torch.as_tensor(value).to(self.device, non_blocking=True)
        if not (isinstance(value, torch.Tensor) and value.is_cuda):
            # the losses select rows by these masks: with the host original at hand they build their index lists without a
            # device sync (losses.host_mask)
            # a PRIVATE copy (N x S bytes): the caller's array may be a staging buffer that is rewritten two batches later
            t._pero_host = np.array(value.numpy() if isinstance(value, torch.Tensor) else value, copy=True)
        return t

    @staticmethod
    def batch_size(batch):
        return len(batch["images"])
