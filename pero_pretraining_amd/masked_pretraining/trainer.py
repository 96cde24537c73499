"""Trainer with the surface of masked_pretraining/trainer.py: train(end, start, view_step),
train_step(batch) -> loss, on_view_step callback.  `step` is an alias of train_step."""
import time

import torch

from ..precision import autocast


class Trainer:
    def __init__(self, batch_operator, model, dataloader, optimizer, scheduler, bfloat16=False, data_parallel=None):
        self.batch_operator = batch_operator
        self.model = model
        self.dataloader = dataloader
        self.optimizer = optimizer
        self.scheduler = scheduler
        self.bfloat16 = bfloat16
        self.data_parallel = data_parallel  # parallel.DataParallel or None
        self.on_view_step = None

    def train(self, end_iteration, start_iteration=0, view_step=1000):
        dataloader_iterator = iter(self.dataloader)
        start_time = time.time()
        iteration_count = 0
        for iteration in range(start_iteration, end_iteration + 1):
            try:
                batch = next(dataloader_iterator)
            except StopIteration:
                dataloader_iterator = iter(self.dataloader)
                batch = next(dataloader_iterator)
            self.scheduler.update_learning_rate(iteration)
            self.train_step(batch)
            # (the reference empties the allocator cache here every iteration, trainer.py:41-42: a perf bug,
            #  deliberately not reproduced)
            iteration_count += 1
            if self.on_view_step is not None and iteration > 0 and iteration % view_step == 0:
                elapsed_time = time.time() - start_time
                self.on_view_step(iteration, self.model, elapsed_time, iteration_count)
                iteration_count = 0
                start_time = time.time()

    def train_step(self, batch):
        images, labels, mask = self.batch_operator.prepare_batch(batch)
        return self.train_step_prepared(images, labels, mask)

    def train_step_prepared(self, images, labels, mask):
        self.optimizer.zero_grad()
        with autocast(self.bfloat16):
            output = self.model.forward(images, labels, mask)
        loss = output["loss"]
        if self.data_parallel is not None:
            self.data_parallel.begin_backward()
        loss.backward()
        if self.data_parallel is not None:
            self.data_parallel.finish_backward()
        self.optimizer.step()
        return loss

    step = train_step
