"""Label production - the step immediately in front of masked pre-training: the
tokenizer's nearest-code indices (VQ-VAE, scripts/produce_vqvae_labels.py:27-46) or nearest-centroid assignments
(k-means, scripts/produce_kmeans_labels.py:31-95) of every line, written as the text file `line_id l1 l2 ...` that
`scripts/convert_gt_to_lmdb.py` turns into the LMDB records the training `DatasetLMDB` reads.

The search runs in the HIP argmin kernel (no (M, K) distance matrix; `models/autoencoders.py`); what travels to the
host per line is the 8-byte index of each valid position.  The encoder that produces the features (a VGG stack in the
reference) is the caller's: `encode(images) -> (N, D, 1, T)` or `(N, D, T)` float tensor on the device."""
import json

import numpy as np
import torch

from ..models.autoencoders import kmeans_labels


# ---- text / record formats ------------------------------------------------------------------------------------------
def save_labels(data, path):
    """scripts/common.py:51-54: one line per image, `line_id` then the labels, space separated (an image without
    labels keeps the trailing space)."""
    with open(path, "w") as f:
        for line_id, line_labels in data.items():
            f.write(f"{line_id} {' '.join([str(label) for label in line_labels])}\n")


def parse_line(line):
    """common/dataset.py:61-69 (Dataset._parse_line): -> (image_id, list of label strings or None)."""
    if " " in line:
        image_id, *labels = line.strip().split()
    else:
        image_id = line.strip()
        labels = None
    return image_id, labels


def label_record(index, image_path, labels):
    """scripts/convert_gt_to_lmdb.py:36: LMDB key / value of line `index`; labels are the file's strings."""
    return f"{index:10d}".encode(), json.dumps({"image": image_path, "labels": labels}).encode()


def parse_label_record(value):
    """common/dataset.py:156-160, 163/171: -> (image id or list of ids, labels)."""
    info = json.loads(value)
    return (info["image"] if "image" in info else info["images"]), info["labels"]


def convert_labels_file(input_path, put, offset=0):
    """scripts/convert_gt_to_lmdb.py:28-39 with the store abstracted to `put(key, value)` (lmdb: `txn.put`).  Lines
    without labels are skipped but still consume an index, as in the reference.  Returns the number of records."""
    n = 0
    with open(input_path, "r") as f:
        for i, line in enumerate(f):
            parts = line.strip().split(" ")
            if not parts[1:]:
                print("Warning: No labels for ", parts[0])
                continue
            put(*label_record(offset + i, parts[0], parts[1:]))
            n += 1
    return n


# ---- label computation ------------------------------------------------------------------------------------------------
def _valid_positions(batch, n, t):
    masks = batch["image_masks"]
    masks = masks.cpu().numpy() if isinstance(masks, torch.Tensor) else np.asarray(masks)
    assert masks.shape == (n, t), (masks.shape, n, t)
    return masks


def compute_labels(encode, quantizer, dataset, batch_operator=None):
    """scripts/produce_vqvae_labels.py:27-46: {line_id: [labels of the positions inside the line]}.
    `quantizer` is `models.autoencoders.VectorQuantizer` (eval mode), `encode` maps the prepared images to (N, D, 1, T)."""
    data = {}
    quantizer.eval()
    with torch.no_grad():
        for batch in dataset:
            images = batch_operator.prepare_batch(batch) if batch_operator is not None else batch["images"]
            tokens, labels = quantizer(encode(images))
            n, _, _, t = tokens.shape
            labels = labels.reshape(n, t).cpu().numpy()
            masks = _valid_positions(batch, n, t)
            for line_id, line_mask, line_labels in zip(batch["ids"], masks, labels):
                data[line_id] = line_labels[line_mask == 1].tolist()
    return data


def compute_kmeans_labels(encode, centroids, dataset, output_path, batch_operator=None):
    """scripts/produce_kmeans_labels.py:31-95: nearest-centroid labels, streamed to `output_path` line by line (same
    text format).  centroids (K, F) device tensor; `encode` -> (N, F, T) or (N, F, 1, T)."""
    count = 0
    with open(output_path, "w") as out, torch.no_grad():
        for batch in dataset:
            images = batch_operator.prepare_batch(batch) if batch_operator is not None else batch["images"]
            features = encode(images)
            if features.dim() == 4:
                features = features.squeeze(2)
            n, _, t = features.shape
            assignment = kmeans_labels(features, centroids).cpu().numpy()
            masks = _valid_positions(batch, n, t)
            for line_id, line_mask, line_ids in zip(batch["ids"], masks, assignment):
                print(line_id, " ".join([str(label) for label in line_ids[line_mask == 1]]), file=out)
                count += 1
    return count
