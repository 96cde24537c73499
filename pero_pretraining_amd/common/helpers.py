"""common/helpers.py of the reference (checkpoint / visualization file names) plus the resume state the reference
omits: optimizer moments, step counter and the host / device RNG streams that draw the masks and the positional
offsets, so that `--start-iteration` continues the SAME trajectory instead of restarting Adam from zero moments."""
import os

import numpy as np
import torch


def get_checkpoint_path(checkpoints_directory, iteration):
    """common/helpers.py:3-4."""
    return os.path.join(checkpoints_directory, f"checkpoint_{iteration:06d}.pth")


def get_visualization_path(visualizations_directory, iteration, part):
    """common/helpers.py:6-7."""
    return os.path.join(visualizations_directory, f"{part}_{iteration:06d}.png")


def get_training_state_path(checkpoints_directory, iteration):
    return os.path.join(checkpoints_directory, f"training_state_{iteration:06d}.pth")


def save_training_state(path, optimizer, iteration, device=None):
    """Everything besides the model weights (which stay in the reference's own checkpoint file / format).  Only tensors,
    numbers, strings, lists and dicts: the file loads with torch.load(weights_only=True)."""
    kind, keys, pos, has_gauss, cached = np.random.get_state()
    state = {"iteration": int(iteration), "optimizer": optimizer.state_dict(),
             "numpy_rng": {"kind": kind, "keys": torch.from_numpy(keys.astype(np.int64)), "pos": int(pos),
                           "has_gauss": int(has_gauss), "cached_gaussian": float(cached)},
             "torch_rng": torch.get_rng_state()}
    if torch.cuda.is_available():
        state["cuda_rng"] = torch.cuda.get_rng_state(device if device is not None else torch.cuda.current_device())
    torch.save(state, path)


def load_training_state(path, optimizer, device=None, restore_rng=True):
    """Returns the iteration the state was saved at."""
    state = torch.load(path, map_location="cpu", weights_only=True)
    optimizer.load_state_dict(state["optimizer"])
    if restore_rng:
        r = state["numpy_rng"]
        np.random.set_state((r["kind"], r["keys"].numpy().astype(np.uint32), r["pos"], r["has_gauss"], r["cached_gaussian"]))
        torch.set_rng_state(state["torch_rng"])
        if "cuda_rng" in state and torch.cuda.is_available():
            torch.cuda.set_rng_state(state["cuda_rng"], device if device is not None else torch.cuda.current_device())
    return state["iteration"]
