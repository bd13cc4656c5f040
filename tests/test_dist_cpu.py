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


def _grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ps = [torch.nn.Parameter(torch.zeros(3, 4)), torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(2)),
          torch.nn.Parameter(torch.zeros(4))]
    ps[0].grad = torch.full((3, 4), float(rank + 1))
    ps[1].grad = torch.arange(5, dtype=torch.float32) * (rank + 1)
    # ps[2]: no gradient on any rank.  ps[3]: only rank 1 produced one (e.g. its step touched another module)
    if rank == 1:
        ps[3].grad = torch.full((4,), 8.0)
    nbytes = sharding.all_reduce_gradients(ps)
    # a rank with NO gradients at all must still enter the collective (the old early return hung the others)
    qs = [torch.nn.Parameter(torch.zeros(6))]
    if rank == 0:
        qs[0].grad = torch.full((6,), 2.0)
    sharding.all_reduce_gradients(qs)
    import random
    rng = random.Random(1234 + 77 * rank)                      # ranks would draw different values on their own
    picks = [sharding.broadcast_choice(1000, rng) for _ in range(5)]
    # plain lists, not tensors: torch ships tensors through a queue as shared-memory FILES, which are gone if this
    # process exits before the parent has opened them (seen once as FileNotFoundError under load)
    q.put((rank, nbytes, ps[0].grad.tolist(), ps[1].grad.tolist(), ps[2].grad, ps[3].grad.tolist(), qs[0].grad.tolist(), picks))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_all_reduce_two_ranks():
    """The fine-tune step's only collective: one flat bucket with a rank-invariant layout, mean over ranks; ranks that
    would otherwise hold gradients of different parameter subsets neither mix them nor hang; per-step random
    choices come from rank 0."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted((q.get(timeout=120) for _ in ps), key=lambda t: t[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, nbytes, g0, g1, g2, g3, h0, picks in res:
        assert nbytes == (12 + 5 + 2 + 4 + 4) * 4             # every parameter + one presence word each
        assert torch.equal(torch.tensor(g0), torch.full((3, 4), 1.5))
        assert torch.equal(torch.tensor(g1), torch.arange(5, dtype=torch.float32) * 1.5)
        assert g2 is None                                       # untouched everywhere: the optimiser skips it
        assert torch.equal(torch.tensor(g3), torch.full((4,), 4.0))           # (0 + 8) / 2 on BOTH ranks, on the right parameter
        assert torch.equal(torch.tensor(h0), torch.full((6,), 1.0))
    assert res[0][7] == res[1][7]                               # the sampled indices agree (rank 0 draws)
    import random
    r0 = random.Random(1234)
    assert res[0][7] == [r0.randint(0, 999) for _ in range(5)]


def test_finetune_host_logic():
    """extract_quality_ref / rems_quality_list / RateLoss mirror training/step.py:14-32, train.py:167-181,
    training/loss.py:189-229."""
    import math
    from vampic import finetune as ft
    assert ft.extract_quality_ref(0.5, [0.75]) is None
    assert ft.extract_quality_ref(2.5, [0.75]) == 0.75
    assert ft.extract_quality_ref(1.0, [0.75, 2.0]) == 0.75 and ft.extract_quality_ref(5.0, [0.75, 2.0]) == 2.0
    assert ft.extract_quality_ref(2.0, [0.5, 1.0, 3.0]) == 1.0 and ft.extract_quality_ref(7.0, [0.5, 1.0, 3.0]) == 3.0
    qs = ft.rems_quality_list([0.75], [10])
    assert qs[0] == 0.76 and qs[-1] == 10 and all(q > 0.75 for q in qs) and len(qs) == 11
    x = torch.rand(2, 3, 8, 8)
    out = {"x_hat": x.clone(), "likelihoods": {"y": torch.full((2, 4, 2, 2), 0.5), "z": torch.full((2, 2, 1, 1), 0.25)}}
    c = ft.RateLoss()(out, x)
    n_pix = 2 * 8 * 8
    assert abs(float(c["bpp_base"]) - 32 * 1.0 / n_pix) < 1e-6 and abs(float(c["bpp_hype"]) - 4 * 2.0 / n_pix) < 1e-6
    assert abs(float(c["loss"]) - (32 + 2 * 8) / n_pix) < 1e-6 and float(c["mse_loss"].mean()) == 0.0


def test_bench_entry_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (how the driver calls it): the parent starts two ranks
    through torch.distributed.run before touching any GPU and relays rank 0's JSON line.  --dry skips the GPU step;
    rank set-up, the barrier-bracketed timing, max-over-ranks and the whole-job rate are the real code."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VAMPIC_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--dry"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                              # ONE line, from rank 0
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 4 and rec["scaling"] == "weak" and rec["config"]["global_batch"] == 64
    assert rec["ms_per_step"] >= 2.0                              # rank 1 sleeps 2 ms per step: the slowest rank sets the time
    assert abs(rec["value"] - 2 * 32 * 65536 * 4 / 1e6 / (rec["ms_per_step"] * 4e-3)) / rec["value"] < 1e-3


def test_bucket_partition_layout():
    """The first-stage gradient buffer is cut into buckets in the order the backward finishes the parameters: disjoint
    cover, every bucket but the last at least the target size, ready steps monotone and closing with the plan."""
    import random
    rng = random.Random(3)
    numels = [rng.choice([3, 32, 192, 36864, 331776, 1128960]) for _ in range(400)]
    offs, tot = [], 0
    for n in numels:
        offs.append(tot)
        tot += (n + 3) // 4 * 4
    done = sorted(rng.randint(1, 5000) for _ in numels)          # finished in layout order ...
    done[37], done[38] = done[38], done[37]                       # ... except one pair (lockstep groups finish together)
    bounds, ready = sharding.bucket_partition(offs, numels, done, tot, 5001, 4 << 20)
    assert bounds[0][0] == 0 and bounds[-1][1] == tot and all(a[1] == b[0] for a, b in zip(bounds, bounds[1:]))
    assert all((hi - lo) * 4 >= (4 << 20) for lo, hi in bounds[:-1])
    assert ready == sorted(ready) and ready[-1] == 5001
    for (lo, hi), r in zip(bounds, ready):                        # a bucket is never sent before its last parameter is final
        assert all(d <= r for o, d in zip(offs, done) if lo <= o < hi)
    assert len(bounds) > 10


def _bucket_worker(rank, world, port, path):
    import json
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(7)
    ps = [torch.nn.Parameter(torch.zeros(n)) for n in (40, 8, 100, 12, 60, 4)]
    offs, tot = [], 0
    for p in ps:
        offs.append(tot)
        tot += (p.numel() + 3) // 4 * 4
    flat = torch.zeros(tot)
    views = [flat[o:o + p.numel()] for o, p in zip(offs, ps)]
    bounds, ready = sharding.bucket_partition(offs, [p.numel() for p in ps], [1, 2, 3, 4, 5, 6], tot, 6, 200)
    red = sharding.BucketReducer()
    # the "backward": parameter k's gradient is written at step k + 1; buckets go out as they become final
    g_local = [torch.randn(p.numel(), generator=torch.Generator().manual_seed(100 * rank + k)) * (3.0 + rank) for k, p in enumerate(ps)]
    nb = 0
    for step in range(1, 7):
        views[step - 1].copy_(g_local[step - 1])
        while nb < len(ready) and ready[nb] <= step:
            lo, hi = bounds[nb]
            red(nb, flat[lo:hi])
            nb += 1
    red.finish()
    for p, v in zip(ps, views):
        p.grad = v.clone()
    norm = float(torch.nn.utils.clip_grad_norm_(ps, 1.0))          # AFTER the exchange: the same scale on every rank
    with open(path, "w") as f:
        json.dump({"rank": rank, "log": red.log, "bounds": bounds, "ready": ready, "norm": norm,
                   "grads": [p.grad.tolist() for p in ps], "local": [g.tolist() for g in g_local]}, f)
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_gradient_exchange_two_ranks(tmp_path):
    """first_train's exchange (BASELINE configs[3]): buckets leave in layout order while the "backward" still runs, every
    rank ends with the MEAN gradient, and clip_grad_norm_ after the last bucket applies one global scale everywhere
    (training/step.py:98 on the averaged gradients = the single-process step on the global batch)."""
    import json
    ctx = mp.get_context("spawn")
    port = _free_port()
    paths = [str(tmp_path / f"r{r}.json") for r in range(2)]
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, paths[r])) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    r0, r1 = (json.load(open(p)) for p in paths)
    assert r0["bounds"] == r1["bounds"] and r0["ready"] == r1["ready"] and len(r0["bounds"]) >= 2
    assert [i for i, _ in r0["log"]] == list(range(len(r0["bounds"]))) == [i for i, _ in r1["log"]]      # issue order
    assert [n for _, n in r0["log"]] == [hi - lo for lo, hi in r0["bounds"]]
    mean = [(torch.tensor(a) + torch.tensor(b)) / 2 for a, b in zip(r0["local"], r1["local"])]
    total = torch.sqrt(sum((m.double() ** 2).sum() for m in mean))
    scale = min(1.0, 1.0 / (float(total) + 1e-6))
    assert abs(r0["norm"] - float(total)) < 1e-4 and r0["norm"] == r1["norm"] and r0["norm"] > 1.0
    for g0, g1, m in zip(r0["grads"], r1["grads"], mean):
        assert g0 == g1                                              # bit-identical on both ranks
        assert torch.allclose(torch.tensor(g0), m * scale, rtol=1e-5, atol=1e-7)


def _forced_worker(port, path):
    import json
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    assert not sharding.collectives_active()
    assert not sharding.init_single_rank_group(None, "gloo")          # flag not set: nothing happens
    os.environ["VAMPIC_FORCE_COLLECTIVES"] = "1"
    assert sharding.init_single_rank_group(None, "gloo") and dist.get_world_size() == 1
    assert sharding.collectives_active()
    flat = torch.arange(24, dtype=torch.float32)
    want = flat.clone()
    red = sharding.BucketReducer()
    red(0, flat[:16])
    red(1, flat[16:])
    n_pending = len(red.pending)
    red.finish()
    p = torch.nn.Parameter(torch.zeros(3))
    p.grad = torch.tensor([1.0, 2.0, 3.0])
    nbytes = sharding.all_reduce_gradients([p])
    os.environ["VAMPIC_FORCE_COLLECTIVES"] = "0"
    off = sharding.collectives_active()
    with open(path, "w") as f:
        json.dump({"pending": n_pending, "same": bool(torch.equal(flat, want)), "log": red.log, "nbytes": nbytes,
                   "grad": p.grad.tolist(), "max": sharding.max_over_ranks(2.5), "off": off}, f)
    dist.destroy_process_group()


def test_forced_collectives_at_world_size_one(tmp_path):
    """VAMPIC_FORCE_COLLECTIVES=1: a 1-rank group issues every exchange step (what the one-GPU box runs over RCCL,
    tests/test_gpu_collectives.py); the values are unchanged (sums over one rank), and without the flag a 1-rank group
    stays silent."""
    import json
    ctx = mp.get_context("spawn")
    path = str(tmp_path / "forced.json")
    p = ctx.Process(target=_forced_worker, args=(_free_port(), path))
    p.start()
    p.join(120)
    assert p.exitcode == 0
    r = json.load(open(path))
    assert r == {"pending": 2, "same": True, "log": [[0, 16], [1, 8]], "nbytes": 16, "grad": [1.0, 2.0, 3.0], "max": 2.5, "off": False}
