"""Trainer with the surface of joint_embedding_pretraining/trainer.py."""
from ..masked_pretraining.trainer import Trainer as _Base
from ..precision import autocast


class Trainer(_Base):
    def train_step(self, batch):
        return self.train_step_prepared(*self.batch_operator.prepare_batch(batch))

    def train_step_prepared(self, images1, images2, image_masks1, image_masks2, shift_masks1, shift_masks2):
        self.optimizer.zero_grad()
        with autocast(self.bfloat16):
            output = self.model.forward(images1, images2, image_masks1, image_masks2, shift_masks1, shift_masks2)
        loss = output["loss"]
        if self.data_parallel is not None:
            # NTXentLoss(cross_rank_negatives=True) applies the upstream gradient once, at the end of its backward - gradients that arrive from
            # other ranks' negatives are scaled by the LOCAL seed, which is right only if every rank seeds the same scalar (here: 1)
            if getattr(self.model.loss, "cross_rank_negatives", False) and getattr(self.data_parallel, "loss_weighting", "rank_mean") != "rank_mean":
                raise ValueError("NTXentLoss(cross_rank_negatives=True) needs the same backward seed on every rank: DataParallel(loss_weighting='rank_mean')")
            self.data_parallel.begin_backward()
        loss.backward()
        if self.data_parallel is not None:
            self.data_parallel.finish_backward()
        self.optimizer.step()
        return loss

    step = train_step
