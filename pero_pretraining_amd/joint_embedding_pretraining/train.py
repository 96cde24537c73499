"""The object wiring of the reference's joint_embedding_pretraining/train.py (init_model, init_batch_operator,
init_testers, init_training, view_step_handler) without its command line, datasets, ClearML and cv2 visualizers; the
resume additions are those of masked_pretraining/train.py (training state file beside the reference-format checkpoint)."""
from functools import partial

from ..common.helpers import get_checkpoint_path, get_training_state_path, save_training_state
from ..common.lr_scheduler import WarmupSchleduler
from ..masked_pretraining.train import resume, save_model  # noqa: F401  (same semantics)
from ..optim import FusedAdam
from .batch_operator import BatchOperator
from .losses import NTXentLoss, VICRegLoss
from .model import JointEmbeddingTransformerEncoder, init_backbone, init_head
from .tester import Tester
from .trainer import Trainer


def init_model(device, backbone_definition, head_definition, loss_type="vicreg", path=None, global_statistics=False):
    """train.py:54-72."""
    backbone = init_backbone(backbone_definition)
    head = init_head(head_definition)
    if loss_type == "vicreg":
        # global_statistics: VICReg mean / variance / covariance over the lines of all data-parallel ranks (losses.py)
        loss = VICRegLoss(global_statistics=global_statistics)
    elif loss_type == "ntxent":
        loss = NTXentLoss()
    else:
        raise ValueError(f"Unknown loss type: {loss_type}")
    model = JointEmbeddingTransformerEncoder(backbone, head, loss)
    model.to(device)
    if path is not None:
        model.load(path)
    return model


def init_batch_operator(device):
    return BatchOperator(device=device)


def init_testers(batch_operator, model, trn_dataloader, tst_dataloader, bfloat16=False):
    """train.py:124-128."""
    return (Tester(batch_operator, model, trn_dataloader, max_lines=1000, bfloat16=bfloat16),
            Tester(batch_operator, model, tst_dataloader, bfloat16=bfloat16))


def report(iteration, dataset, result, scheduler, clearml_logger=None):
    """train.py:157-166."""
    name = dataset.name() if callable(getattr(dataset, "name", None)) else str(getattr(dataset, "name", "dataset"))
    print(f"TEST {name} iteration:{iteration} loss:{float(result['loss']):.6f} lr:{scheduler.current_lr:.6e}")
    if clearml_logger is not None:
        clearml_logger.report_scalar(title="loss", series=name, value=result["loss"], iteration=iteration)


def test_model(iteration, tester, scheduler, clearml_logger=None):
    result = tester.test()
    report(iteration, tester.dataloader, result, scheduler, clearml_logger=clearml_logger)
    return result


def view_step_handler(iteration, model, elapsed_time, iteration_count, trn_tester, tst_tester, checkpoints_directory,
                      scheduler, optimizer=None, clearml_logger=None):
    print(f"Iteration: {iteration}, time: {elapsed_time:.2f} s, speed: {iteration_count / elapsed_time:.2f} it/s.")
    save_model(model, get_checkpoint_path(checkpoints_directory, iteration))
    for tester in (trn_tester, tst_tester):
        if tester is not None:
            test_model(iteration, tester, scheduler, clearml_logger=clearml_logger)
    if optimizer is not None:
        save_training_state(get_training_state_path(checkpoints_directory, iteration), optimizer, iteration)


def init_training(batch_operator, model, dataset, trn_tester, tst_tester, learning_rate, warmup_iterations,
                  checkpoints_directory, bfloat16=False, clearml_logger=None, data_parallel=None):
    """train.py:131-148 with FusedAdam in place of torch.optim.Adam."""
    optimizer = FusedAdam(model.parameters(), lr=learning_rate)
    scheduler = WarmupSchleduler(optimizer, learning_rate, warmup_iterations, 1)
    trainer = Trainer(batch_operator, model, dataset, optimizer, scheduler, bfloat16=bfloat16, data_parallel=data_parallel)
    trainer.on_view_step = partial(view_step_handler, trn_tester=trn_tester, tst_tester=tst_tester,
                                   checkpoints_directory=checkpoints_directory, scheduler=scheduler, optimizer=optimizer,
                                   clearml_logger=clearml_logger)
    return trainer
