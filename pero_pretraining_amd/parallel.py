"""Data parallelism for the pre-training step: one process per GPU, `torch.distributed` (backend
"nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for tests), gradient all-reduce of the optimizer's flat
f32 gradient buffer.

The reference is single-process: lines are independent through the whole encoder, so the
minibatch is sharded by line and the only exchange is the gradient reduction.  Design for xGMI
(point-to-point links, ring collectives are per-link bound): few large collectives - one bucket per
encoder layer (12.6 MB f32 at d=512) carved out of the ONE flat gradient buffer, launched as soon as that
layer's backward kernels are enqueued (the backbone calls `_on_layer_grads_ready(i)`), so communication
of layer i overlaps the backward compute of layers i-1..0.  The sum is turned into a mean by the fused
Adam kernel's `grad_scale` (no extra pass over the gradients).

Loss semantics under sharding (stated, as SURVEY.md section 8e asks).  loss_weighting="rank_mean" (default): each
rank's loss is the mean over ITS masked positions and gradients are averaged over ranks (mean of per-rank means -
identical to torch DistributedDataParallel; equals the single-process gradient when every rank has the same number of
masked positions).  loss_weighting="global_mean": the masked counts are all-reduced (8 bytes, on the device, no host
sync) and rank r's backward is seeded with n_r * world / n_global, so the averaged gradient IS the gradient of the mean
over all masked positions of the global batch - exactly what the single-process reference computes on that batch.
VICReg statistics are per rank.
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world_size):
    """Contiguous shard [begin, end) of n_items for `rank` (first n_items % world ranks get one more)."""
    base, rem = divmod(n_items, world_size)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def plan_buckets(named_params, offsets, num_layers, layer_prefix="backbone.encoder_layers.layers."):
    """Group the flat-buffer ranges of the parameters by the backward stage that completes them.

    named_params: iterable of (name, param); offsets: {id(param): (group, start, numel)} (padded layout of
    optim.FusedAdam).  Returns {stage: [(group, start, end), ...]} with stage = layer index for encoder
    layers, -1 for the rest of the backbone (front end; finished last) and "head" for everything else
    (finished before the backbone's backward starts).  Adjacent ranges are merged."""
    stages = {}
    for name, p in named_params:
        if id(p) not in offsets:
            continue
        g, start, n = offsets[id(p)]
        if name.startswith(layer_prefix):
            stage = int(name[len(layer_prefix):].split(".")[0])
        elif name.startswith("backbone."):
            stage = -1
        else:
            stage = "head"
        stages.setdefault(stage, []).append((g, start, start + ((n + 7) // 8) * 8))
    merged = {}
    for stage, ranges in stages.items():
        ranges.sort()
        out = [list(ranges[0])]
        for g, a, b in ranges[1:]:
            if g == out[-1][0] and a <= out[-1][2]:
                out[-1][2] = max(out[-1][2], b)
            else:
                out.append([g, a, b])
        merged[stage] = [tuple(r) for r in out]
    return merged


class DataParallel:
    """Gradient reduction driver.  Usage (see masked_pretraining/trainer.py):
        dp = DataParallel(model, optimizer); trainer = Trainer(..., data_parallel=dp)
    """

    def __init__(self, model, optimizer, process_group=None, overlap=True, layers_per_bucket=2, loss_weighting="rank_mean"):
        if loss_weighting not in ("rank_mean", "global_mean"):
            raise ValueError(f"Unknown loss weighting: {loss_weighting}")
        self.loss_weighting = loss_weighting
        self.model, self.optimizer, self.group = model, optimizer, process_group
        self.layers_per_bucket = max(1, int(layers_per_bucket))  # 2 layers = 25 MB f32 at d=512
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = dist.is_initialized()  # a 1-rank group still exercises the collective path
        self.overlap = overlap
        self._pending = []
        self._done = set()
        backbone = getattr(model, "backbone", None)
        self.num_layers = len(backbone.encoder_layers.layers) if backbone is not None else 0
        self.flat = optimizer.flat_grads()
        self.buckets = plan_buckets(model.named_parameters(), optimizer.param_offsets(), self.num_layers)
        optimizer.grad_scale = 1.0 / self.world_size
        if backbone is not None:
            backbone._on_layer_grads_ready = self._stage_ready if overlap else None
        self.broadcast_parameters()

    def broadcast_parameters(self):
        """Start every rank from rank 0's weights."""
        if not self.active:
            return
        for f in self.optimizer._flat:
            if f is not None:
                dist.broadcast(f["p"], src=0, group=self.group)
        self.optimizer.refresh_lowp()

    def _reduce(self, stage):
        if stage in self._done or stage not in self.buckets:
            return
        self._done.add(stage)
        for g, a, b in self.buckets[stage]:
            self._pending.append(dist.all_reduce(self.flat[g][a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _stage_ready(self, stage):
        """Called by the backbone's backward when layer `stage` (or -1: the front end) has enqueued its last
        gradient kernel.  Layers are reduced in groups of `layers_per_bucket` (fewer, larger collectives: xGMI ring
        collectives are per-link bound and every launch costs a few microseconds of CU time)."""
        if not self.active or getattr(self, "_defer", False):
            return
        self._reduce("head")   # complete before the backbone's backward began
        if stage == -1:
            self._reduce(-1)
            return
        lpb = self.layers_per_bucket
        if stage % lpb == 0:  # lowest layer of its group: the whole group [stage, stage + lpb) is complete
            for s in range(stage, min(stage + lpb, self.num_layers)):
                self._reduce(s)

    def backward_seed(self, local_count):
        """d(loss)/d(loss) seed for this rank's backward: None (= 1) for rank_mean, n_r * world / n_global for
        global_mean.  local_count: 0-dim tensor (number of positions in this rank's mean), stays on its device."""
        if self.loss_weighting == "rank_mean" or not self.active:
            return None
        total = local_count.detach().to(torch.float32).clone()
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=self.group)
        return local_count.detach().to(torch.float32) * float(self.world_size) / total

    def begin_backward(self):
        """Call after the forward pass(es), before loss.backward().  The bucket hooks assume ONE backbone backward per step: a
        step that ran the backbone forward more than once (the joint model's non-batched fallback) would fire every stage in
        the first of its backward passes and add the second one's gradients to buckets already reduced - such a step defers
        all reduction to finish_backward()."""
        self._pending, self._done = [], set()
        backbone = getattr(self.model, "backbone", None)
        passes = getattr(backbone, "_grad_forwards", 1) if backbone is not None else 1
        if backbone is not None:
            backbone._grad_forwards = 0
        self._defer = passes > 1

    def finish_backward(self):
        """Reduce whatever has not been launched yet and make the compute stream wait for all of it."""
        if not self.active:
            return
        for stage in list(self.buckets.keys()):
            self._reduce(stage)
        for w in self._pending:
            w.wait()
        self._pending = []


def all_reduce_scalar_mean(value, group=None):
    """Mean of a python float over ranks (logging only)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64)
    if dist.get_backend(group) == "nccl":
        t = t.cuda()
    dist.all_reduce(t, group=group)
    return float(t[0]) / dist.get_world_size(group)
