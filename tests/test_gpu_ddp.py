"""GPU: the data-parallel training step with world_size 2 on ONE device (both ranks share cuda:0, gloo backend with
device tensors - NCCL/RCCL needs one device per rank, which the test box does not have).  Checks the real model,
FusedAdam flat buffers, the per-layer bucket hooks fired from the backbone's backward (on the side stream), gradient
averaging via grad_scale, and that both ranks hold identical parameters after the step."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _build():
    from pero_pretraining_amd.masked_pretraining import model as M
    torch.manual_seed(0)
    bb = M.init_backbone({"type": "vit", "num_blocks": 2, "model_dim": 64, "num_heads": 4, "feedforward_dim": 128})
    hd = M.init_head({"type": "linear", "in_features": 64, "out_features": 96})
    return M.MaskedTransformerEncoder(bb, hd).cuda().train()


def _data(rank):
    rng = np.random.default_rng(100 + rank)
    images = rng.integers(0, 256, (3, 40, 128, 3), dtype=np.uint8)
    labels = rng.integers(0, 96, (3, 16)).astype(np.int64)
    mask = np.zeros((3, 16), np.int64)
    mask[:, rank::3] = 1
    return torch.from_numpy(images).cuda(), torch.from_numpy(labels).cuda(), torch.from_numpy(mask).cuda()


def _worker(rank, world, port, out_q, weighting="rank_mean"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pero_pretraining_amd.common.lr_scheduler import WarmupSchleduler
    from pero_pretraining_amd.masked_pretraining.trainer import Trainer
    from pero_pretraining_amd.optim import FusedAdam
    from pero_pretraining_amd.parallel import DataParallel
    model = _build()
    if rank == 1:  # different start: the constructor's broadcast must equalise
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.1)
    opt = FusedAdam(model.parameters(), lr=1e-3)
    dp = DataParallel(model, opt, loss_weighting=weighting)
    trainer = Trainer(None, model, None, opt, WarmupSchleduler(opt, 1e-3, 0, 1), data_parallel=dp)
    offs = np.array([5, 17, 300]) + rank
    model.backbone.set_offsets(offs)
    images, labels, mask = _data(rank)
    # forward/backward without the optimizer step to inspect the reduced gradients
    loss = trainer._forward_backward(images, labels, mask)
    torch.cuda.synchronize()
    grads = {k: (p.grad * opt.grad_scale).cpu().numpy().copy() for k, p in model.named_parameters()}
    opt.step()
    torch.cuda.synchronize()
    params = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
    gathered = [torch.empty_like(params) for _ in range(world)]
    dist.all_gather(gathered, params)
    same = all(torch.equal(gathered[0], g) for g in gathered)
    if rank == 0:
        out_q.put((float(loss), grads, same))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_rank_step_matches_mean_of_single_rank_gradients():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for port in [_free_port()] for r in range(2)]
    for p in procs:
        p.start()
    loss0, grads, same = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert same, "ranks diverged after the optimizer step"
    # reference: each rank's gradient computed alone in this process, averaged
    ref = None
    for rank in range(2):
        model = _build()
        model.backbone.set_offsets(np.array([5, 17, 300]) + rank)
        images, labels, mask = _data(rank)
        res = model(images, labels, mask)
        res["loss"].backward()
        if rank == 0:
            assert abs(float(res["loss"]) - loss0) < 1e-6 * abs(loss0)
        g = {k: p.grad.cpu().numpy() for k, p in model.named_parameters()}
        ref = g if ref is None else {k: ref[k] + g[k] for k in g}
    for k in ref:
        want = ref[k] / 2
        assert np.abs(grads[k] - want).max() <= 1e-5 * max(np.abs(want).max(), 1e-3), k


def test_two_rank_global_mean_equals_single_process_on_the_whole_batch():
    """loss_weighting="global_mean": ranks hold different numbers of masked positions (18 and 15); the averaged gradient
    must be the gradient of ONE mean over all 33 - what the single-process reference computes on the 6-line batch."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, "global_mean")) for r in range(2)]
    for p in procs:
        p.start()
    _, grads, same = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert same
    model = _build()
    model.backbone.set_offsets(np.concatenate([np.array([5, 17, 300]) + r for r in range(2)]))
    parts = [_data(r) for r in range(2)]
    images, labels, mask = (torch.cat([p[i] for p in parts]) for i in range(3))
    assert int(parts[0][2].sum()) != int(parts[1][2].sum())
    model(images, labels, mask)["loss"].backward()
    for k, p in model.named_parameters():
        want = p.grad.cpu().numpy()
        assert np.abs(grads[k] - want).max() <= 1e-5 * max(np.abs(want).max(), 1e-3), k


# ---- exact-global VICReg statistics -------------------------------------------------------
def _vicreg_data(rank):
    """Ragged shard: rank 0 holds 3 lines, rank 1 holds 2; three-valued shift masks, image masks with padding."""
    rng = np.random.default_rng(7 + rank)
    n, s, d = (3, 2)[rank], 24, 256
    x = rng.standard_normal((n, s, d)).astype(np.float32)
    y = (x + 0.3 * rng.standard_normal((n, s, d))).astype(np.float32)
    im1 = np.ones((n, s), np.uint8); im2 = np.ones((n, s), np.uint8)
    sm1 = np.zeros((n, s), np.uint8); sm2 = np.zeros((n, s), np.uint8)
    for i in range(n):
        width = int(rng.integers(12, s + 1))
        im1[i, width:] = 0
        im2[i, width:] = 0
        shift = int(rng.integers(0, 5))
        sm1[i, shift:width] = 1
        sm2[i, :width - shift] = 1
        sm1[i, width:] = 2  # shared padding positions (dataloader.py:137-138): selected by neither loss term
        sm2[i, width:] = 2
    return x, y, im1, im2, sm1, sm2


def _vicreg_worker(rank, world, port, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pero_pretraining_amd.joint_embedding_pretraining.losses import VICRegLoss
    x, y, *masks = _vicreg_data(rank)
    xg = torch.from_numpy(x).cuda().requires_grad_(True)
    yg = torch.from_numpy(y).cuda().requires_grad_(True)
    res = VICRegLoss(variance_weight=2.0, invariance_weight=3.0, covariance_weight=0.5, global_statistics=True)(
        xg, yg, *[torch.from_numpy(m).cuda() for m in masks])
    res["loss"].backward()
    torch.cuda.synchronize()
    out_q.put((rank, {k: float(v) for k, v in res.items()}, xg.grad.cpu().numpy(), yg.grad.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_global_vicreg_statistics_equal_the_loss_on_the_concatenated_batch():
    from oracle import pero_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_vicreg_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict((r, rest) for r, *rest in (q.get(timeout=240) for _ in range(2)))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    parts = [_vicreg_data(r) for r in range(2)]
    x, y, *masks = (np.concatenate([p[i] for p in parts]) for i in range(6))
    xo = torch.from_numpy(x).requires_grad_(True)
    yo = torch.from_numpy(y).requires_grad_(True)
    ref = O.vicreg_loss(xo, yo, *masks, variance_weight=2.0, invariance_weight=3.0, covariance_weight=0.5)
    ref["loss"].backward()
    n0 = parts[0][0].shape[0]
    for r in range(2):
        vals, dx, dy = got[r]
        for k in ref:  # every rank reports the loss of the WHOLE batch
            assert abs(vals[k] - float(ref[k])) <= 1e-4 * abs(float(ref[k])) + 1e-7, (r, k, vals[k], float(ref[k]))
        rows = slice(0, n0) if r == 0 else slice(n0, None)
        for have, want in ((dx, xo.grad[rows].numpy()), (dy, yo.grad[rows].numpy())):
            # seeded with world_size: the data-parallel average (divide by 2) gives the single-process gradient
            assert np.abs(have / 2 - want).max() <= 1e-4 * np.abs(want).max()


# ---------------------------------------------------------------------------------------------------------------------
# NT-Xent with cross-rank negatives (extension; BASELINE.json configs[4]): two ranks == the oracle on the concatenated batch
def _ntxent_cross_worker(rank, world, port, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pero_pretraining_amd.joint_embedding_pretraining.losses import NTXentLoss
    rng = np.random.default_rng(7)
    n, s, D = 3, 16, 64
    xa = rng.standard_normal((world * n, s, D)).astype(np.float32)
    ya = (xa + 0.5 * rng.standard_normal((world * n, s, D))).astype(np.float32)
    x = torch.from_numpy(xa[rank * n:(rank + 1) * n]).cuda().requires_grad_(True)
    y = torch.from_numpy(ya[rank * n:(rank + 1) * n]).cuda().requires_grad_(True)
    ones = np.ones((n, s), np.uint8)
    loss = NTXentLoss(cross_rank_negatives=True)(x, y, ones, ones, ones, ones)["loss"]
    loss.backward()
    torch.cuda.synchronize()
    out_q.put((rank, float(loss), x.grad.cpu().numpy(), y.grad.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_ntxent_cross_rank_negatives_two_ranks_equal_the_oracle_on_the_concatenated_batch():
    from oracle import pero_oracle as O
    world, n, s, D = 2, 3, 16, 64
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ntxent_cross_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    rng = np.random.default_rng(7)
    xa = rng.standard_normal((world * n, s, D)).astype(np.float32)
    ya = (xa + 0.5 * rng.standard_normal((world * n, s, D))).astype(np.float32)
    xo = torch.from_numpy(xa).double().requires_grad_(True)
    yo = torch.from_numpy(ya).double().requires_grad_(True)
    mean, per_rank = O.ntxent_cross_loss(xo, yo, n)
    (per_rank.sum()).backward()        # every rank differentiates ITS loss; gradients through the gathered rows are summed back
    for r, loss, gx, gy in res:
        assert abs(loss - float(per_rank[r])) < 1e-4 * abs(float(per_rank[r])), (r, loss, float(per_rank[r]))
        rx, ry = xo.grad[r * n:(r + 1) * n].numpy(), yo.grad[r * n:(r + 1) * n].numpy()
        assert np.abs(gx - rx).max() < 1e-4 * np.abs(rx).max() + 1e-8, r
        assert np.abs(gy - ry).max() < 1e-4 * np.abs(ry).max() + 1e-8, r
    assert abs(sum(t[1] for t in res) / world - float(mean)) < 1e-4 * float(mean)


# ---------------------------------------------------------------------------------------------------------------------
# the joint-embedding Trainer under data parallelism (two ranks, one device, gloo)
def _joint_dp_worker(rank, world, port, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pero_pretraining_amd.common.lr_scheduler import WarmupSchleduler
    from pero_pretraining_amd.joint_embedding_pretraining.batch_operator import BatchOperator
    from pero_pretraining_amd.joint_embedding_pretraining.losses import VICRegLoss
    from pero_pretraining_amd.joint_embedding_pretraining.model import JointEmbeddingTransformerEncoder, LinearHead
    from pero_pretraining_amd.joint_embedding_pretraining.trainer import Trainer
    from pero_pretraining_amd.models.transformers import VisionTransformerEncoder
    from pero_pretraining_amd.optim import FusedAdam
    from pero_pretraining_amd.parallel import DataParallel
    torch.manual_seed(3)
    bb = VisionTransformerEncoder(num_blocks=2, model_dim=64, num_heads=4, feedforward_dim=128)
    model = JointEmbeddingTransformerEncoder(bb, LinearHead(in_features=64, out_features=80), VICRegLoss()).cuda().train()
    opt = FusedAdam(model.parameters(), lr=1e-3)
    dp = DataParallel(model, opt)
    trainer = Trainer(BatchOperator(torch.device("cuda", 0)), model, None, opt, WarmupSchleduler(opt, 1e-3, 0, 1), data_parallel=dp)
    rng = np.random.default_rng(50 + rank)
    n, S = 3, 24
    ones = np.ones((n, S), np.uint8)
    sm = ones.copy()
    sm[:, :2] = 0
    batch = {"images": rng.integers(0, 256, (n, 40, S * 8, 3), dtype=np.uint8), "images2": rng.integers(0, 256, (n, 40, S * 8, 3), dtype=np.uint8),
             "image_masks": ones, "image_masks2": ones, "shift_masks": sm, "shift_masks2": sm[:, ::-1].copy()}
    model.backbone.set_offsets(np.arange(n) + 7 * rank, np.arange(n) + 100 + rank)
    loss = trainer.train_step(batch)
    torch.cuda.synchronize()
    params = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
    gathered = [torch.empty_like(params) for _ in range(world)]
    dist.all_gather(gathered, params)
    out_q.put((rank, float(loss), all(torch.equal(gathered[0], g) for g in gathered), bool(torch.isfinite(params).all())))
    dist.barrier()
    dist.destroy_process_group()


def test_joint_trainer_two_rank_data_parallel_step_keeps_the_ranks_identical():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_joint_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(same and finite and np.isfinite(loss) for _, loss, same, finite in res)
    assert abs(res[0][1] - res[1][1]) > 0     # different shards, different losses


# ---------------------------------------------------------------------------------------------------------------------
# The RCCL-specific branches (all_gather_into_tensor / reduce_scatter_tensor of the cross-rank NT-Xent, the bucketed all-reduce of
# DataParallel on its comm stream, the global VICReg exchanges) with a ONE-rank RCCL process group on the real device: the collectives
# are identities, so every result must equal the group-less computation - what is tested is that RCCL accepts the buffers (dtype,
# contiguity, sizes, streams) the code hands it.  More ranks need more GPUs than the test box has.
def _rccl_one_rank_worker(port, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from pero_pretraining_amd.joint_embedding_pretraining.losses import NTXentLoss, VICRegLoss
    res = {"backend": dist.get_backend()}
    rng = np.random.default_rng(7)
    n, s, D = 3, 16, 64
    xa = rng.standard_normal((n, s, D)).astype(np.float32)
    ya = (xa + 0.5 * rng.standard_normal((n, s, D))).astype(np.float32)
    ones = np.ones((n, s), np.uint8)
    for tag, lossmod in (("ntxent_group", NTXentLoss(cross_rank_negatives=True)),):
        x = torch.from_numpy(xa).cuda().requires_grad_(True)
        y = torch.from_numpy(ya).cuda().requires_grad_(True)
        loss = lossmod(x, y, ones, ones, ones, ones)["loss"]
        loss.backward()
        torch.cuda.synchronize()
        res[tag] = (float(loss), x.grad.cpu().numpy(), y.grad.cpu().numpy())
    xv, yv, *masks = _vicreg_data(0)
    for tag, glob in (("vicreg_global", True), ("vicreg_local", False)):
        xg = torch.from_numpy(xv).cuda().requires_grad_(True)
        yg = torch.from_numpy(yv).cuda().requires_grad_(True)
        r = VICRegLoss(global_statistics=glob)(xg, yg, *[torch.from_numpy(m).cuda() for m in masks])
        r["loss"].backward()
        torch.cuda.synchronize()
        res[tag] = (float(r["loss"]), xg.grad.cpu().numpy(), yg.grad.cpu().numpy())
    # the masked step under DataParallel (bucket hooks -> RCCL all-reduce on the comm stream) against the same step without it
    from pero_pretraining_amd.common.lr_scheduler import WarmupSchleduler
    from pero_pretraining_amd.masked_pretraining.trainer import Trainer
    from pero_pretraining_amd.optim import FusedAdam
    from pero_pretraining_amd.parallel import DataParallel
    for tag, use_dp in (("step_dp", True), ("step_plain", False)):
        model = _build()
        opt = FusedAdam(model.parameters(), lr=1e-3)
        dp = DataParallel(model, opt) if use_dp else None
        trainer = Trainer(None, model, None, opt, WarmupSchleduler(opt, 1e-3, 0, 1), data_parallel=dp)
        model.backbone.set_offsets(np.array([5, 17, 300]))
        images, labels, mask = _data(0)
        loss = trainer.train_step_prepared(images, labels, mask)
        torch.cuda.synchronize()
        res[tag] = (float(loss), torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu().numpy())
    out_q.put(res)
    dist.barrier()
    dist.destroy_process_group()


def test_one_rank_rccl_group_runs_every_collective_path_and_changes_nothing():
    from oracle import pero_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_one_rank_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert res["backend"] == "nccl"
    # cross-rank NT-Xent with one rank == the oracle's definition on that batch
    rng = np.random.default_rng(7)
    n, s, D = 3, 16, 64
    xa = rng.standard_normal((n, s, D)).astype(np.float32)
    ya = (xa + 0.5 * rng.standard_normal((n, s, D))).astype(np.float32)
    xo = torch.from_numpy(xa).double().requires_grad_(True)
    yo = torch.from_numpy(ya).double().requires_grad_(True)
    mean, per_rank = O.ntxent_cross_loss(xo, yo, n)
    per_rank.sum().backward()
    loss, gx, gy = res["ntxent_group"]
    assert abs(loss - float(mean)) < 1e-4 * float(mean)
    assert np.abs(gx - xo.grad.numpy()).max() < 1e-4 * np.abs(xo.grad.numpy()).max() + 1e-8
    assert np.abs(gy - yo.grad.numpy()).max() < 1e-4 * np.abs(yo.grad.numpy()).max() + 1e-8
    # global VICReg statistics over one rank == the local statistics
    (lg, xg, yg), (ll, xl, yl) = res["vicreg_global"], res["vicreg_local"]
    assert abs(lg - ll) <= 1e-5 * abs(ll)
    assert np.abs(xg - xl).max() <= 1e-4 * np.abs(xl).max() and np.abs(yg - yl).max() <= 1e-4 * np.abs(yl).max()
    # the data-parallel step with one rank == the plain step
    (ld, pd), (lp, pp) = res["step_dp"], res["step_plain"]
    assert abs(ld - lp) <= 1e-6 * abs(lp)
    assert np.abs(pd - pp).max() <= 1e-6


def test_bench_two_ranks_rehearsal_emits_the_world_gt_1_legs():
    """`python bench.py --gpus 2` on THIS box: the self-launch relay starts two ranks that share the one GPU and exchange over gloo
    (PERO_BENCH_REHEARSE_GLOO=1 - RCCL needs a device per rank), i.e. every world > 1 code path of the bench runs: the data-parallel
    masked step, config 4 under DataParallel with per-rank and with exact global VICReg statistics, config 5 with cross-rank negatives.
    The timings mean nothing (two ranks on one device) and the line says so; what is checked is that the ONE JSON line arrives with
    world_size 2 in the headline and in each of the three legs, finite losses, and n_gpus == 2."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PERO_BENCH_REHEARSE_GLOO="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--repeats", "1", "--batch", "32",
                        "--leg-pairs", "8,16", "--no-cpu-baseline", "--no-sweep", "--no-options", "--no-roofline"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["distributed"]["world_size"] == 2 and "REHEARSAL" in d["distributed"]["backend"] and d["distributed"]["self_launched"]
    assert d["config"]["global_batch"] == 64 and np.isfinite(d["final_loss"]) and d["value"] > 0
    want = {"config4_vicreg_dp", "config4_vicreg_dp_global_statistics", "config5_ntxent_dp_cross_rank_negatives"}
    assert set(d["legs"]) == want, set(d["legs"])
    for k in want:
        leg = d["legs"][k]
        assert leg["distributed"]["world_size"] == 2 and np.isfinite(leg["loss"]) and leg["ms_per_step"] > 0, (k, leg)
