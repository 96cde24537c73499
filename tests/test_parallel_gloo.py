"""CPU, world_size 2 over gloo: the data-parallel driver (shard ranges, bucket plan, bucketed all-reduce of the
flat gradient buffer, parameter broadcast, mean via the optimizer's grad_scale)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pero_pretraining_amd.parallel import DataParallel, plan_buckets, shard_range


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 64, 4097):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= 1


class _Layer(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(5, 3))
        self.b = torch.nn.Parameter(torch.zeros(5))


class _Backbone(torch.nn.Module):
    def __init__(self, n):
        super().__init__()
        self.encoder_layers = torch.nn.Module()
        self.encoder_layers.layers = torch.nn.ModuleList([_Layer() for _ in range(n)])
        self.conv = torch.nn.Parameter(torch.zeros(7))
        self._on_layer_grads_ready = None


class _Model(torch.nn.Module):
    def __init__(self, n=3):
        super().__init__()
        self.backbone = _Backbone(n)
        self.head = torch.nn.Linear(3, 2)


class _FlatOpt:
    """CPU stand-in with the flat-buffer surface of optim.FusedAdam."""

    def __init__(self, model):
        ps = list(model.parameters())
        offs, total = [], 0
        for p in ps:
            offs.append(total)
            total += ((p.numel() + 7) // 8) * 8
        self.p, self.g = torch.zeros(total), torch.zeros(total)
        for p, o in zip(ps, offs):
            self.p[o:o + p.numel()].copy_(p.detach().reshape(-1))
            p.data = self.p[o:o + p.numel()].view(p.shape)
            p.grad = self.g[o:o + p.numel()].view(p.shape)
        self._flat = [None, dict(p=self.p, g=self.g)]   # group 0 holds no parameters (a frozen / empty group): indices must not shift
        self.ps, self.offs, self.grad_scale, self.refreshed = ps, offs, 1.0, 0

    def flat_grads(self):
        return {1: self.g}

    def param_offsets(self):
        return {id(p): (1, o, p.numel()) for p, o in zip(self.ps, self.offs)}

    def refresh_lowp(self):
        self.refreshed += 1


def test_bucket_plan_stages():
    model = _Model(3)
    opt = _FlatOpt(model)
    plan = plan_buckets(model.named_parameters(), opt.param_offsets(), 3)
    assert set(plan) == {0, 1, 2, -1, "head"}
    covered = sorted(r for v in plan.values() for r in v)
    assert covered[0][1] == 0 and covered[-1][2] == opt.g.numel()
    assert all(a[2] == b[1] for a, b in zip(covered, covered[1:]))  # disjoint and complete
    assert all(len(plan[i]) == 1 and plan[i][0][2] - plan[i][0][1] == 16 + 8 for i in range(3))


def _worker(rank, world, port, overlap):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)  # different initial weights per rank: broadcast must equalise them
    model = _Model(3)
    with torch.no_grad():
        for p in model.parameters():
            p.normal_()
    opt = _FlatOpt(model)
    dp = DataParallel(model, opt, overlap=overlap)
    ref = [torch.empty_like(opt.p) for _ in range(world)]
    dist.all_gather(ref, opt.p)
    assert all(torch.equal(ref[0], r) for r in ref) and opt.refreshed == 1
    assert opt.grad_scale == 1.0 / world
    # "backward": every rank fills its gradients with rank-specific values, stage by stage
    dp.begin_backward()
    for i, p in enumerate(model.parameters()):
        p.grad.fill_(float(rank + 1) * (i + 1))
    if overlap:
        for stage in (2, 1, 0, -1):
            model.backbone._on_layer_grads_ready(stage)
    dp.finish_backward()
    for i, p in enumerate(model.parameters()):
        expect = sum(float(r + 1) * (i + 1) for r in range(world))
        assert torch.all(p.grad == expect), (i, p.grad, expect)
        assert torch.all(p.grad * opt.grad_scale == expect / world)
    # a step with TWO backbone backward passes (the joint model's non-batched fallback): the hooks of the first pass must
    # not reduce anything - every gradient is complete only after the second - and finish_backward reduces all of it once
    if overlap:
        model.backbone._grad_forwards = 2
        dp.begin_backward()
        assert model.backbone._grad_forwards == 0
        for i, p in enumerate(model.parameters()):
            p.grad.fill_(float(rank + 1))
        for stage in (2, 1, 0, -1):
            model.backbone._on_layer_grads_ready(stage)          # first pass
        assert not dp._pending and not dp._done
        for i, p in enumerate(model.parameters()):
            p.grad.add_(float(rank + 1) * i)                      # second pass accumulates
        for stage in (2, 1, 0, -1):
            model.backbone._on_layer_grads_ready(stage)
        dp.finish_backward()
        for i, p in enumerate(model.parameters()):
            assert torch.all(p.grad == sum(float(r + 1) * (1 + i) for r in range(world))), i
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_bucketed_all_reduce_world2_overlap():
    mp.spawn(_worker, args=(2, _free_port(), True), nprocs=2, join=True)


def test_all_reduce_world2_no_overlap():
    mp.spawn(_worker, args=(2, _free_port(), False), nprocs=2, join=True)


def _seed_worker(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = _Model(2)
    counts = [3, 5]
    dp = DataParallel(model, _FlatOpt(model), loss_weighting="global_mean")
    seed = dp.backward_seed(torch.tensor(counts[rank]))
    assert abs(float(seed) - counts[rank] * world / sum(counts)) < 1e-7     # 0.75 / 1.25: mean of seeds over ranks == 1
    assert DataParallel(model, _FlatOpt(model)).backward_seed(torch.tensor(counts[rank])) is None
    dist.destroy_process_group()


def test_global_mean_backward_seed_world2():
    mp.spawn(_seed_worker, args=(2, _free_port()), nprocs=2, join=True)


def test_bench_self_launch_relays_one_json_line_world2():
    """`python bench.py --gpus 2` with no RANK in the environment starts the two ranks itself (a torch.distributed.run child; the
    parent never touches a GPU) and relays rank 0's ONE JSON line: rehearsed on CPU over gloo with the GPU work replaced by a
    trivial step (--plumbing-test): rank environment, process group, barrier-bracketed max-over-ranks timing, descriptor juggling."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--plumbing-test", "--steps", "4", "--warmup", "1",
                        "--repeats", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=240)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 1 and d["repeats"] == 2
    assert d["distributed"] == {"world_size": 2, "backend": "gloo", "self_launched": True}
    assert d["all_reduce_check"] == 2.0
    assert d["ms_per_step"] >= 2.0          # the slowest rank (2 ms sleep per step) sets the time: max over ranks
    # --spawn: the same path with one rank
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--spawn", "--plumbing-test", "--steps", "2", "--warmup", "0",
                        "--repeats", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=240)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = json.loads(r.stdout.decode().strip())
    assert d["distributed"] == {"world_size": 1, "backend": "gloo", "self_launched": True}
    # a mismatch between --gpus and the launcher's world size is an error, not a silent single-rank run
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--plumbing-test"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, env=env2, timeout=120)
    assert r.returncode != 0 and b"WORLD_SIZE" in r.stderr
