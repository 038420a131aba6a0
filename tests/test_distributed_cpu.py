"""world_size-2 gloo test of the N>1 plumbing used by bench.py (image sharding, barrier,
max-over-ranks clock).  Runs on CPU."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from casapose_amd import parallel

    r, l, w = parallel.init_from_env("gloo")
    assert (r, w) == (rank, world)
    b, e = parallel.shard_range(37, r, w)
    parallel.barrier_sync()
    mx = parallel.max_over_ranks(1.0 + rank)          # the slowest rank's clock
    total = parallel.sum_over_ranks(float(e - b))     # images processed by all ranks
    q.put((rank, b, e, mx, total))
    torch.distributed.destroy_process_group()


def test_two_rank_sharding_and_clock():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, b0, e0, m0, t0), (r1, b1, e1, m1, t1) = res
    assert (b0, e0, b1, e1) == (0, 19, 19, 37)   # ragged split, union = range(37)
    assert m0 == m1 == 2.0 and t0 == t1 == 37.0


def test_shard_range_properties():
    from casapose_amd.parallel import shard_range

    for total in (0, 1, 7, 16, 257):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _syncbn_worker(rank, world, port, q):
    """The SyncBN protocol of the training step on CPU tensors: every rank holds half of the batch, all-reduces
    the fp64 statistic tables exactly as TrainPlan does, and must reproduce the whole-batch normalisation and its
    gradient (compared with the autograd oracle on the full batch)."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import numpy as np

    from casapose_amd import parallel

    parallel.init_from_env("gloo")
    rng = np.random.default_rng(0)                      # same data on both ranks; each takes its shard
    n, c = 64, 8
    x = torch.from_numpy(rng.standard_normal((n, c)) * 2 + 1)
    dy = torch.from_numpy(rng.standard_normal((n, c)))
    gamma = torch.from_numpy(1 + 0.1 * rng.standard_normal(c))
    b, e = parallel.shard_range(n, rank, world)
    xl, dyl = x[b:e], dy[b:e]
    eps = 2e-5
    sums = torch.cat([xl.sum(0), (xl * xl).sum(0)])    # cp_bn_stats_f32
    parallel.all_reduce_sum_(sums, None, world)
    N = (e - b) * world
    mean = sums[:c] / N
    var = sums[c:] / N - mean * mean
    rstd = torch.rsqrt(var + eps)
    xh = (xl - mean) * rstd
    g = dyl * gamma
    chan = torch.cat([g.sum(0), (g * xh).sum(0)])       # cp_bn_act_bwd_reduce_f32 (chan table)
    parallel.all_reduce_sum_(chan, None, world)
    dx = rstd * (g - chan[:c] / N - xh * chan[c:] / N)  # cp_bn_act_bwd_apply_f32
    grad = torch.cat([(dyl * xh).sum(0)])               # d gamma, local part
    parallel.all_reduce_sum_(grad, None, world)         # the flat-gradient all-reduce
    q.put((rank, b, e, mean.numpy(), var.numpy(), dx.numpy(), grad.numpy()))
    torch.distributed.destroy_process_group()


def test_two_rank_syncbn_protocol_matches_whole_batch():
    import numpy as np

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_syncbn_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(0)
    n, c = 64, 8
    x = torch.tensor(rng.standard_normal((n, c)) * 2 + 1, requires_grad=True)
    dy = torch.from_numpy(rng.standard_normal((n, c)))
    gamma = torch.tensor(1 + 0.1 * rng.standard_normal(c), requires_grad=True)
    mean = x.mean(0)
    var = ((x - mean) ** 2).mean(0)
    y = (x - mean) / torch.sqrt(var + 2e-5) * gamma
    y.backward(dy)
    dx = np.concatenate([r[5] for r in res])
    assert np.allclose(res[0][3], mean.detach().numpy(), atol=1e-12) and np.allclose(res[1][4], var.detach().numpy(), atol=1e-10)
    assert np.allclose(dx, x.grad.numpy(), atol=1e-10)
    assert np.allclose(res[0][6], gamma.grad.numpy(), atol=1e-10) and np.allclose(res[1][6], gamma.grad.numpy(), atol=1e-10)


def _log_worker(rank, world, port, q):
    import numpy as np

    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from casapose_amd import parallel
    from casapose_amd.data_handler.synthetic_scene import SyntheticSceneDataset

    parallel.init_from_env("gloo")
    calls = {"n": 0}
    real = torch.distributed.all_reduce

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)

    torch.distributed.all_reduce = counting
    losses = [1.0 + rank, 2.0 * rank, 0.5, 3.0, rank * rank]
    stats = [np.arange(8, dtype=np.float64) * (j + 1) + rank for j in range(8)]
    mean, st = parallel.reduce_step_log(losses, stats, world, None)
    mean2, st2 = parallel.reduce_step_log(losses, None, world, None)
    torch.distributed.all_reduce = real
    # every replica renders only its slice of the GLOBAL batch (ragged: 7 images over 3 replicas), and the slices partition it
    ds = SyntheticSceneDataset(3, (64, 64), length=7, seed=5, random_crop=False)
    it, n = ds.generate_dataset(7, 1, shard=(rank, world))
    mine = next(it)
    q.put((rank, mean, st, mean2, st2 is None, calls["n"], mine["img"].numpy(), n))
    torch.distributed.destroy_process_group()


def test_three_rank_packed_logging_and_source_sharding():
    """ONE collective per step for the logged quantities (5 losses as MEAN, 6 pose-statistic vectors as SUM; train_casapose.py:690-694,
    732-737) and per-replica data sharding at the source, world size 3 with a ragged global batch."""
    import numpy as np

    from casapose_amd.data_handler.synthetic_scene import SyntheticSceneDataset

    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_log_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_mean = [np.mean([1.0 + r for r in range(3)]), np.mean([2.0 * r for r in range(3)]), 0.5, 3.0, np.mean([r * r for r in range(3)])]
    want_st = np.stack([sum(np.arange(8, dtype=np.float64) * (j + 1) + r for r in range(3)) for j in range(6)])
    whole = SyntheticSceneDataset(3, (64, 64), length=7, seed=5, random_crop=False).batch(0, 7)["img"].numpy()
    got = np.concatenate([r[6] for r in res])
    assert [r[6].shape[0] for r in res] == [3, 2, 2] and np.array_equal(got, whole)
    for rank, mean, st, mean2, none2, ncalls, _, n in res:
        assert np.allclose(mean, want_mean) and np.allclose(mean2, want_mean) and np.allclose(st, want_st) and none2
        assert ncalls == 2 and n == 1                     # exactly one all-reduce per logged step
