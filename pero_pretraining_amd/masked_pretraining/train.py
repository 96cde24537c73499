"""The object wiring of the reference's masked_pretraining/train.py (init_model, init_batch_operator, init_testers,
init_training, report, test_model, save_model, view_step_handler) - everything between the command line and
`Trainer.train`, without the command line itself, datasets, ClearML and the cv2 visualizers.

Differences, all on the resume path: the optimizer is `FusedAdam`; `view_step_handler` also writes
`training_state_{it:06d}.pth` (optimizer moments + RNG streams) beside the reference-format `checkpoint_{it:06d}.pth`,
and `resume()` restores both, so `--start-iteration N` continues the interrupted trajectory (the reference reloads the
weights only and restarts Adam from zero moments)."""
import os
from functools import partial

import torch

from ..common.helpers import (get_checkpoint_path, get_training_state_path, load_training_state, save_training_state)
from ..common.lr_scheduler import WarmupSchleduler
from ..optim import FusedAdam
from .batch_operator import BatchOperator
from .model import MaskedCrossEntropyLoss, MaskedTransformerEncoder, init_backbone, init_head
from .tester import Tester
from .trainer import Trainer


def init_model(device, backbone_definition, head_definition, path=None, unmasked_weight=None):
    """train.py:62-74."""
    backbone = init_backbone(backbone_definition)
    head = init_head(head_definition)
    loss = MaskedCrossEntropyLoss(unmasked_weight=unmasked_weight)
    model = MaskedTransformerEncoder(backbone, head, loss=loss)
    model.to(device)
    if path is not None:
        model.load(path)
    return model


def init_batch_operator(device, masking_prob):
    return BatchOperator(device=device, masking_prob=masking_prob)


def init_testers(batch_operator, model, trn_dataloader, tst_dataloader, bfloat16=False):
    """train.py:137-141."""
    return (Tester(batch_operator, model, trn_dataloader, max_lines=1000, bfloat16=bfloat16),
            Tester(batch_operator, model, tst_dataloader, bfloat16=bfloat16))


def report(iteration, dataset, result, scheduler, clearml_logger=None):
    """train.py:168-190 (same line format)."""
    errors_keys = sorted([key for key in result.keys() if key.startswith("errors_")], key=lambda key: int(key.split("_")[-1]))
    name = dataset.name() if callable(getattr(dataset, "name", None)) else str(getattr(dataset, "name", "dataset"))
    print(f"TEST {name} iteration:{iteration} loss:{float(result['loss']):.6f} "
          f"errors:{'|'.join(str(result[k]) for k in errors_keys)} lr:{scheduler.current_lr:.6e}")
    if clearml_logger is not None:
        clearml_logger.report_scalar(title="loss", series=name, value=result["loss"], iteration=iteration)
        for k in errors_keys:
            clearml_logger.report_scalar(title=f"top-{k.split('_')[-1]}_error", series=name, value=result[k], iteration=iteration)


def test_model(iteration, tester, scheduler, clearml_logger=None):
    result = tester.test()
    report(iteration, tester.dataloader, result, scheduler, clearml_logger=clearml_logger)
    return result


def save_model(model, path):
    model.save(path)


def view_step_handler(iteration, model, elapsed_time, iteration_count, trn_tester, tst_tester, checkpoints_directory,
                      scheduler, optimizer=None, trn_visualizer=None, tst_visualizer=None, visualizations_directory=None,
                      clearml_logger=None):
    """train.py:207-217; visualizers are optional (they need cv2), the training state file is new."""
    print(f"Iteration: {iteration}, time: {elapsed_time:.2f} s, speed: {iteration_count / elapsed_time:.2f} it/s.")
    save_model(model, get_checkpoint_path(checkpoints_directory, iteration))
    if trn_tester is not None:
        test_model(iteration, trn_tester, scheduler, clearml_logger=clearml_logger)
    if tst_tester is not None:
        test_model(iteration, tst_tester, scheduler, clearml_logger=clearml_logger)
    if optimizer is not None:  # last: the testers draw masks from the same host RNG stream the trainer continues with
        save_training_state(get_training_state_path(checkpoints_directory, iteration), optimizer, iteration)


def init_training(batch_operator, model, dataset, trn_tester, tst_tester, learning_rate, warmup_iterations,
                  checkpoints_directory, bfloat16=False, clearml_logger=None, data_parallel=None):
    """train.py:144-163 with FusedAdam in place of torch.optim.Adam (same hyper-parameters)."""
    optimizer = FusedAdam(model.parameters(), lr=learning_rate)
    scheduler = WarmupSchleduler(optimizer, learning_rate, warmup_iterations, 1)
    trainer = Trainer(batch_operator, model, dataset, optimizer, scheduler, bfloat16=bfloat16, data_parallel=data_parallel)
    trainer.on_view_step = partial(view_step_handler, trn_tester=trn_tester, tst_tester=tst_tester,
                                   checkpoints_directory=checkpoints_directory, scheduler=scheduler, optimizer=optimizer,
                                   clearml_logger=clearml_logger)
    return trainer


def resume(trainer, checkpoints_directory, start_iteration):
    """Restore what `view_step_handler` wrote at `start_iteration` (weights: reference-format file; optimizer + RNG: the
    training state file when present).  Returns the iteration to pass to `Trainer.train(start_iteration=...)`:
    N + 1 when the full state was restored (the trajectory continues), N otherwise - the reference's behaviour, which
    repeats iteration N on freshly zeroed Adam moments (train.py:243-251 + trainer.py:26)."""
    trainer.model.load(get_checkpoint_path(checkpoints_directory, start_iteration))
    state_path = get_training_state_path(checkpoints_directory, start_iteration)
    if os.path.exists(state_path):
        load_training_state(state_path, trainer.optimizer)
        return start_iteration + 1
    return start_iteration
