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
