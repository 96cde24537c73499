"""Trainer with the surface of masked_pretraining/trainer.py: train(end, start, view_step),
train_step(batch) -> loss, on_view_step callback.  `step` is an alias of train_step."""
import time

import torch

from ..precision import autocast


class _StepGraph:
    """zero_grad -> forward -> backward of one step captured as a hipGraph (all kernels, both streams) and replayed
    with one launch: removes ~600 per-kernel host launches (6-7 ms of host time per step) and the gaps between
    kernels.  Inputs are copied into static buffers; the optimizer step stays outside (its learning rate and
    bias corrections are host scalars that change every step).  One graph per input-shape signature."""

    def __init__(self, trainer, tensors):
        self.static_in = [t.clone() for t in tensors]
        self.trainer = trainer
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):  # warm-up outside capture (allocator, lazy inits, lowp caches)
            for _ in range(2):
                self.loss = trainer._forward_backward(*self.static_in)
        torch.cuda.current_stream().wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = trainer._forward_backward(*self.static_in)

    def replay(self, tensors):
        for dst, src in zip(self.static_in, tensors):
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.loss


class Trainer:
    def __init__(self, batch_operator, model, dataloader, optimizer, scheduler, bfloat16=False, data_parallel=None,
                 hip_graph=False):
        if hip_graph:
            # the captured step replays fixed launches: the bf16 weight refresh must be inside the graph (FusedAdam's own
            # kernel; torch optimizers would leave the forward on the weights of capture time) and nothing may read the
            # mask on the host during capture (head_rows == "masked" lists the rows with torch.nonzero)
            from ..optim import FusedAdam
            if not isinstance(optimizer, FusedAdam):
                raise ValueError("Trainer(hip_graph=True) needs optim.FusedAdam")
            if getattr(model, "head_rows", "all") != "all":
                raise ValueError("Trainer(hip_graph=True) needs model.head_rows == 'all'")
        self.hip_graph = hip_graph
        self._graphs = {}
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

    def _forward_backward(self, images, labels, mask, rows=None):
        self.optimizer.zero_grad()
        with autocast(self.bfloat16):
            output = self.model.forward(images, labels, mask) if rows is None else self.model.forward(images, labels, mask, rows=rows)
        loss = output["loss"]
        seed = None
        if self.data_parallel is not None:
            if self.data_parallel.loss_weighting == "global_mean":
                if getattr(self.model.loss, "unmasked_weight", None) is not None:
                    raise ValueError("loss_weighting='global_mean' covers the masked mean only (unmasked_weight must be None)")
                m = mask if isinstance(mask, torch.Tensor) else torch.as_tensor(mask)
                seed = self.data_parallel.backward_seed((m.to(loss.device) == 1).sum())
            self.data_parallel.begin_backward()
        loss.backward(seed)
        if self.data_parallel is not None:
            self.data_parallel.finish_backward()
        return loss.detach()

    def train_step_prepared(self, images, labels, mask, rows=None):
        """rows (optional, model.head_rows == "masked"): int64 device tensor of the flat positions with mask == 1."""
        if self.hip_graph and self.data_parallel is None and rows is None:
            tensors = [images, torch.as_tensor(labels).to(images.device), torch.as_tensor(mask).to(images.device)]
            key = tuple((tuple(t.shape), t.dtype) for t in tensors) + (self.model.training,)
            g = self._graphs.get(key)
            if g is None:
                g = self._graphs[key] = _StepGraph(self, tensors)
            loss = g.replay(tensors)
        else:
            loss = self._forward_backward(images, labels, mask, rows)
        self.optimizer.step()
        return loss

    step = train_step
