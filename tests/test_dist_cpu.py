"""N > 1 path on CPU: two gloo ranks shard a job of images, aggregate the slowest rank's time
and per-image scalars exactly like bench.py does over RCCL."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vampic import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, e = sharding.shard_range(7, rank, world)
    dt = sharding.max_over_ranks(0.5 + rank)                     # rank 1 is the slow one
    tot = sharding.sum_over_ranks([float(e - b), float(sum(range(b, e)))])
    dist.barrier()
    q.put((rank, b, e, dt, tot))
    dist.destroy_process_group()


def test_two_rank_sharding_and_aggregation():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    (r0, b0, e0, dt0, tot0), (r1, b1, e1, dt1, tot1) = res
    assert (b0, e0, b1, e1) == (0, 4, 4, 7)                     # disjoint cover, balanced
    assert dt0 == dt1 == 1.5                                    # max over ranks
    assert tot0 == tot1 == [7.0, 21.0]                          # every image counted once
    v = sharding.whole_job_megapixels_per_s(32, 256, 256, 10, 2, 1.5)
    assert abs(v - 2 * 32 * 65536 * 10 / 1e6 / 1.5) < 1e-9


def test_shard_range_properties():
    for n in (0, 1, 7, 32, 257):
        for w in (1, 2, 3, 8):
            spans = [sharding.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
