"""common/dataloader.py of the reference (BatchCreator: the collate function that turns ragged text lines into one
padded batch), MI355X-native.

The reference builds the batch on the host: zero-filled (B, H, Wt, 3) numpy arrays, one slice copy per line and view,
python loops for the masks, then `BatchOperator` uploads the padded arrays.  Here the host only takes the decisions that
consume its RNG stream - the random left paddings and crop offsets, drawn by the same `np.random.randint` calls in the same
order, so a seeded run collates identically - and uploads the ragged pixels once, back to back (sum(w) * H * 3 bytes
instead of B * Wt * H * 3); `pero_stack_lines` writes the padded uint8 batch in HBM and `pero_line_masks` the image masks,
shifts and three-valued shift masks.  The batch dict has the reference's keys; `images*` / `*_masks*` are uint8 DEVICE
tensors (the batch operators take tensors as well as numpy arrays), `labels` stays a host numpy array because
`BatchOperator._create_mask` draws the masking pattern from it on the host like the reference.
"""
from typing import Dict, List

import numpy as np
import torch

from .. import ops


def create_dataloader(dataset, batch_creator=None, batch_size=16, shuffle=False, num_workers=0, persistent_workers=True,
                      drop_last=True):
    """common/dataloader.py:6-20.  The collate function launches HIP kernels, so it runs in the training process
    (num_workers must stay 0: worker processes cannot share the parent's HIP context); decoding workers belong in the
    dataset, in front of it."""
    from torch.utils.data import DataLoader
    if num_workers != 0:
        raise ValueError("BatchCreator collates on the GPU: use num_workers=0 for the DataLoader that owns it")
    if batch_creator is None:
        batch_creator = BatchCreator()
    return DataLoader(dataset, batch_size=batch_size, shuffle=shuffle, num_workers=0, collate_fn=batch_creator.create_batch,
                      drop_last=drop_last)


class BatchCreator:
    def __init__(self, subsampling_factor=8, padding_coefficient=32, crop_width=None, crop_step=1, same_left_paddings=False,
                 device=None):
        self.subsampling_factor = subsampling_factor
        self.padding_coefficient = padding_coefficient
        self.crop_width = crop_width
        self.crop_step = crop_step
        self.same_left_paddings = same_left_paddings
        self.device = torch.device("cuda") if device is None else torch.device(device)

    # ---- reference surface -------------------------------------------------------------------------------------
    def create_batch(self, data: List[Dict]) -> Dict:
        """dataloader.py:32-60."""
        (images, images2, image_masks, image_masks2, left_paddings, left_paddings2, original_images, original_images2,
         shifts, shift_masks1, shift_masks2) = self.stack_images(data)
        target_labels_length = images.shape[2] // self.subsampling_factor
        labels, ids = self.stack_annotations(data, target_labels_length, left_paddings)
        return {"images": images, "images2": images2, "image_masks": image_masks, "image_masks2": image_masks2,
                "shifts": shifts, "shift_masks": shift_masks1, "shift_masks2": shift_masks2, "labels": labels, "ids": ids,
                "original_images": original_images, "original_images2": original_images2}

    def stack_annotations(self, data: List[Dict], target_labels_length, left_paddings):
        """dataloader.py:62-73 (host: a few hundred integers per line, consumed by the host-side mask draw)."""
        ids = [d["image_id"] for d in data]
        labels = None
        if any(d["labels"] is not None for d in data):
            labels = np.full((len(data), target_labels_length), fill_value=-1)
            for i, (d, lp) in enumerate(zip(data, left_paddings)):
                if d["labels"] is not None:
                    labels[i, lp:lp + len(d["labels"])] = d["labels"]
        return labels, ids

    def stack_images(self, data: List[Dict]):
        """dataloader.py:75-155.  Same 11-tuple; images / masks are device tensors, paddings and shifts python lists."""
        if self.crop_width is not None:
            crop_shifts = self.crop_images(data)
            target_width = self.crop_width
        else:
            crop_shifts = [0] * len(data)
            all_widths = [d["image"].shape[1] for d in data] + [d["image2"].shape[1] for d in data
                                                                if "image2" in d and d["image2"] is not None]
            target_width = self.calculate_padded_image_width(max(all_widths))
        sub = self.subsampling_factor
        H, C = data[0]["image"].shape[0], data[0]["image"].shape[2]
        paired = any(d["image2"] is not None for d in data)

        lines1 = [d["image"] for d in data]
        left1 = [self._draw_left_padding(l, target_width) for l in lines1]           # RNG: view 1 first ...
        lines2 = left2 = None
        if paired:
            lines2 = [d["image2"] for d in data]
            left2 = list(left1) if self.same_left_paddings else [self._draw_left_padding(l, target_width) for l in lines2]  # ... then view 2

        images1, w1, l1 = self._stack(lines1, left1, H, target_width, C)
        images2 = None
        w2 = l2 = cs = None
        if paired:
            images2, w2, l2 = self._stack(lines2, left2, H, target_width, C)
            cs = torch.tensor(crop_shifts, dtype=torch.int32).to(self.device, non_blocking=True)
        im1, im2, sm1, sm2, shifts_t = ops.line_masks(w1, l1, target_width // sub, sub, w2, l2, cs)
        shifts = [c + (a - b) for c, a, b in zip(crop_shifts, left1, left2)] if paired else None   # dataloader.py:126
        # The widths and paddings behind these masks are host values: the same few hundred bytes per line are also written on the
        # host and travel with the device masks (`_pero_host`), so that the losses and the masked head list their rows without a
        # device sync (ops.host_mask) - a device-only mask costs a torch.nonzero round trip per mask and step
        host = self._host_masks([l.shape[1] for l in lines1], left1, [l.shape[1] for l in lines2] if paired else None, left2,
                                crop_shifts, target_width // sub, sub)
        for t, h in zip((im1, im2, sm1, sm2), host):
            if t is not None:
                t._pero_host = h

        original_images1 = self._stack_originals(data, "image_original", H, C)
        original_images2 = self._stack_originals(data, "image2_original", H, C)
        return (images1, images2, im1, im2, left1, left2, original_images1, original_images2, shifts, sm1, sm2)

    def crop_images(self, data: List[Dict]):
        """dataloader.py:157-183 (views into the decoded lines, no pixel copies; draws from np.random like the reference)."""
        shifts = []
        sub = self.subsampling_factor
        for d in data:
            d["image_original"] = d["image"]
            d["image2_original"] = d["image2"]
            d["image"], start = self.crop_image(d["image"])
            min_shift = -min(start // sub, self.crop_width // sub - 1)
            max_shift = max(0, min((d["image_original"].shape[1] - start - self.crop_width) // sub, self.crop_width // sub - 1))
            shift = min_shift if min_shift == max_shift else np.random.randint(min_shift, max_shift)
            start += shift * sub
            d["image2"], _ = self.crop_image(d["image2"], start=start)
            shifts.append(shift)
        return shifts

    def crop_image(self, image, start=None):
        """dataloader.py:185-195."""
        if image.shape[1] <= self.crop_width:
            return image, 0
        if start is None:
            diff = image.shape[1] - self.crop_width
            start = np.random.randint(0, diff) // self.crop_step
            start *= self.crop_step
        return image[:, start:start + self.crop_width, :], start

    def calculate_padded_image_width(self, image_width: int):
        """dataloader.py:197-198."""
        return int(np.ceil(image_width / self.padding_coefficient) * self.padding_coefficient) + self.padding_coefficient

    @staticmethod
    def _host_masks(widths1, left1, widths2, left2, crop_shifts, S, sub):
        """The masks of pero_line_masks (csrc/collate.hip; the reference's dataloader.py:92-96, 124-138) on the host: image masks,
        three-valued shift masks (1 shared, 2 shared-but-padding, 0 not shared).  Returns (im1, im2, sm1, sm2), uint8 (B, S)."""
        pos = np.arange(S)[None, :]
        a1 = np.asarray(left1, dtype=np.int64)[:, None]
        e1 = a1 + (np.asarray(widths1, dtype=np.int64)[:, None] + sub - 1) // sub
        im1 = ((pos >= a1) & (pos < e1)).astype(np.uint8)
        if widths2 is None:
            return im1, None, None, None
        a2 = np.asarray(left2, dtype=np.int64)[:, None]
        e2 = a2 + (np.asarray(widths2, dtype=np.int64)[:, None] + sub - 1) // sub
        im2 = ((pos >= a2) & (pos < e2)).astype(np.uint8)
        shift = np.asarray(crop_shifts, dtype=np.int64)[:, None] + a1 - a2
        on1 = np.where(shift < 0, pos < S + shift, pos >= shift)
        rev = S - 1 - pos                                          # shift_mask2 = reverse(shift_mask1)
        on2 = np.where(shift < 0, rev < S + shift, rev >= shift)
        sm1 = np.where(on1, np.where(im1 == 1, 1, 2), 0).astype(np.uint8)
        sm2 = np.where(on2, np.where(im2 == 1, 1, 2), 0).astype(np.uint8)
        return im1, im2, sm1, sm2

    # ---- device side ---------------------------------------------------------------------------------------------
    def _draw_left_padding(self, line_image, target_width):
        """dataloader.py:86-89: in label positions."""
        if line_image.shape[1] == target_width:
            return 0
        return np.random.randint(0, target_width - line_image.shape[1]) // self.subsampling_factor

    def _stack(self, lines, left, H, target_width, C):
        """Upload the ragged lines once (pinned staging buffer) and pad them on the device."""
        sizes = [int(l.shape[0]) * int(l.shape[1]) * int(l.shape[2]) for l in lines]
        total = int(sum(sizes))
        staging = torch.empty(total + 8, dtype=torch.uint8, pin_memory=True)
        host = staging.numpy()
        offsets, pos = [], 0
        for l, n in zip(lines, sizes):
            if l.shape[0] != H or l.shape[2] != C or l.dtype != np.uint8:
                raise ValueError("BatchCreator: every line must be uint8 (H, w, C) with the batch's H and C")
            if l.shape[1] > target_width:
                raise ValueError("BatchCreator: a line is wider than the target width")
            host[pos:pos + n] = np.ascontiguousarray(l).reshape(-1)
            offsets.append(pos)
            pos += n
        host[total:] = 0
        dev = self.device
        packed = staging.to(dev, non_blocking=True)
        meta = torch.tensor([offsets, [l.shape[1] for l in lines], [lp * self.subsampling_factor for lp in left], left],
                            dtype=torch.int64).to(dev, non_blocking=True)
        widths, left_px, left_pos = (meta[i].to(torch.int32) for i in (1, 2, 3))
        images = ops.stack_lines(packed, meta[0].contiguous(), widths, left_px, len(lines), H, target_width, C)
        return images, widths, left_pos

    @staticmethod
    def _stack_originals(data, key, H, C):
        """dataloader.py:140-152: uncropped lines for the visualizers only (crop mode) - host arrays like the reference."""
        if not any(key in d and d[key] is not None for d in data):
            return None
        max_width = max(d[key].shape[1] for d in data)
        out = np.zeros([len(data), H, max_width, C], dtype=np.uint8)
        for row, d in zip(out, data):
            row[:, :d[key].shape[1]] = d[key]
        return out


class DevicePrefetcher:
    """Overlaps `batch_operator.prepare_batch` (host mask draw + host -> device copies) of batch i + 1 with the training step of
    batch i.  The reference's Trainer.train_step (masked_pretraining/trainer.py:53-54) moves each batch synchronously with
    `.to(self.device)`; here the numpy arrays of the next batch are staged in PINNED host buffers (two sets) and copied on a
    dedicated copy stream, and the compute stream only waits for that copy's event - at 245 760 B per 40x2048 line the uint8 batch
    of 1024 lines is ~5 ms of PCIe time under an ~90 ms step.

        for images, labels, mask in DevicePrefetcher(dataloader, BatchOperator(device, 0.15), device):
            trainer.train_step_prepared(images, labels, mask)

    Yields exactly what `prepare_batch` returns (device tensors stay valid until the prefetcher is two batches further)."""

    def __init__(self, batches, batch_operator, device, depth=2):
        self.batches, self.batch_operator, self.device, self.depth = batches, batch_operator, torch.device(device), max(2, int(depth))
        self._pinned = [dict() for _ in range(self.depth)]
        self._done = [None] * self.depth

    def _stage(self, batch, slot, stream):
        if self._done[slot] is not None:
            self._done[slot].synchronize()   # the copy that last read this pinned set (two batches ago: long finished)
        staged = {}
        for k, v in batch.items():
            if isinstance(v, np.ndarray) and v.dtype != object and v.size > 0:
                buf = self._pinned[slot].get(k)
                if buf is None or buf.shape != v.shape or buf.dtype != torch.from_numpy(v[:0]).dtype:
                    buf = self._pinned[slot][k] = torch.empty(v.shape, dtype=torch.from_numpy(v[:0]).dtype, pin_memory=True)
                buf.numpy()[...] = v
                staged[k] = buf
            else:
                staged[k] = v
        with torch.cuda.stream(stream):
            out = self.batch_operator.prepare_batch(staged)
            if isinstance(out, (tuple, list)):
                # host arrays among the results (the masked step's numpy mask): through the pinned set too, so that the model's
                # own `.to(device)` of a pageable array does not stall the launching thread inside the step
                conv = []
                for j, o in enumerate(out):
                    if isinstance(o, np.ndarray) and o.dtype != object:
                        key = ("out", j)
                        buf = self._pinned[slot].get(key)
                        if buf is None or buf.shape != o.shape or buf.dtype != torch.from_numpy(o[:0]).dtype:
                            buf = self._pinned[slot][key] = torch.empty(o.shape, dtype=torch.from_numpy(o[:0]).dtype, pin_memory=True)
                        buf.numpy()[...] = o
                        t = buf.to(self.device, non_blocking=True)
                        t._pero_host = np.array(o, copy=True)   # private: `o` may alias a buffer the producer reuses
                        conv.append(t)
                    else:
                        conv.append(o)
                out = tuple(conv)
            ev = torch.cuda.Event()
            ev.record(stream)
        self._done[slot] = ev
        return out, ev

    def __iter__(self):
        stream = torch.cuda.Stream(self.device)
        it = iter(self.batches)
        try:
            nxt = self._stage(next(it), 0, stream)
        except StopIteration:
            return
        i = 0
        while nxt is not None:
            cur = nxt
            i += 1
            try:
                nxt = self._stage(next(it), i % self.depth, stream)
            except StopIteration:
                nxt = None
            out, ev = cur
            main = torch.cuda.current_stream(self.device)
            main.wait_event(ev)
            for t in (out if isinstance(out, (tuple, list)) else (out,)):
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(main)
            yield out
