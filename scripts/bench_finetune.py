#!/usr/bin/env python3
"""Secondary benchmark: REM fine-tune step (BASELINE.json configs[4]: `--training_type rems`,
check_levels 0.75, batch 128 over 8 GPUs = 16 images of 256x256 per GPU).

A step = checkpoint latent at q=0.75 (no grad) + training-mode forward at q + RateLoss + backward
(HIP kernels) + gradient all-reduce (RCCL, one flat bucket) + clip + Adam — what
training/step.py:56-95 does per batch.  One process per GPU; launched like bench.py:

    python scripts/bench_finetune.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        scripts/bench_finetune.py --gpus N ...

Prints one JSON line: whole-job images/s (weak scaling) and the per-phase time split.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def measure(dev, rank=0, world=1, dist=None, batch=16, size=256, steps=10, warmup=3, quality=2.5, no_graph=False, two_pass=False,
            forced=False, freeze_gc=True):
    """Time ``steps`` REM fine-tune steps on this rank's synthetic shard; returns the record rank 0 prints (bench.py embeds
    it in its own line as ``train.rem_finetune``)."""
    import vampic
    from vampic import finetune as ft, sharding
    args = argparse.Namespace(model="rem", check_levels=[0.75], mu_std=True, dimension="middle", N=192, M=640,
                              multiple_decoder=True, multiple_encoder=True, multiple_hyperprior=True, dim_chunk=32,
                              division_dimension=[320, 640], mask_policy="point-based-std", support_progressive_slices=5,
                              delta_encode=True, total_mu_rep=True, all_scalable=True)
    net = vampic.get_model(args, "cpu")
    torch.nn.Module.load_state_dict(net, vampic.synth.synth_state_dict(net.state_dict(), seed=0))
    net = net.to(dev).train()
    net.freeze_all()
    net.unfreeze_rems()
    net.use_graph = not no_graph
    opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=1e-4)
    crit = ft.RateLoss()
    x = vampic.synth.synth_image(batch, size, size, seed=200 + rank).to(dev)

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(max(warmup, 1)):
        c = ft.finetune_step(net, crit, x, opt, quality, [0.75], fused=not two_pass)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        c = ft.finetune_step(net, crit, x, opt, quality, [0.75], fused=not two_pass)
    sync()
    dt_unfrozen = time.perf_counter() - t0
    # see scripts/bench_train.py: the headline is measured after gc.freeze(); the un-frozen time of the same steps
    # was taken just above and is reported beside it
    import gc
    if freeze_gc:
        gc.collect()
        gc.freeze()
    t0 = time.perf_counter()
    for _ in range(steps):
        c = ft.finetune_step(net, crit, x, opt, quality, [0.75], fused=not two_pass)
    sync()
    dt = sharding.max_over_ranks(time.perf_counter() - t0, dev if (dist is None or dist.get_backend() == "nccl") else "cpu")

    # phase split on rank 0 (events on the current stream; the plans join it on entry and exit)
    def timed(fn):
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        r = fn()
        torch.cuda.synchronize(dev)
        return r, (time.perf_counter() - t) * 1e3
    with torch.no_grad():
        ck, t_ck = timed(lambda: net.ExtractChekpointRepr(x, quality=0.75, rc=False))
    opt.zero_grad()
    if two_pass:
        out, t_fwd = timed(lambda: net.forward_single_quality(x, quality=quality, training=True, checkpoint_ref=ck))
    else:
        t_ck = 0.0
        out, t_fwd = timed(lambda: net.forward_finetune(x, quality))
    loss = crit(out, x)["loss"]
    _, t_bwd = timed(loss.backward)
    _, t_ar = timed(lambda: sharding.all_reduce_gradients(p for p in net.parameters() if p.requires_grad))
    _, t_opt = timed(lambda: (torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0), opt.step()))
    n_par = sum(p.numel() for p in net.parameters() if p.requires_grad)
    gc.unfreeze()
    return ({"metric": "REM fine-tune images/sec (256x256 patches, rate loss, Adam)",
                          "value": round(world * batch * steps / dt, 2), "unit": "images/s", "n_gpus": world,
                          "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 3), "ms_per_step_gc_unfrozen": round(dt_unfrozen / steps * 1e3, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                          "data": "synthetic", "collectives": ("forced (1-rank nccl group)" if forced else ("nccl" if world > 1 else "none")),
                          "config": {"workload": f"REM fine-tune step q={quality}, check level 0.75, {batch}x3x{size}x{size} per GPU",
                                     "global_batch": batch * world, "trainable_params": n_par,
                                     "grad_bucket_bytes": 4 * n_par, "hip_graph": not no_graph, "fused_checkpoint": not two_pass,
                                     "loss": round(float(c["loss"].detach()), 5)},
                          "phase_ms": {"checkpoint_forward": round(t_ck, 3), "train_forward": round(t_fwd, 3),
                                       "backward": round(t_bwd, 3), "grad_all_reduce": round(t_ar, 3),
                                       "clip_adam": round(t_opt, 3)}})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--quality", type=float, default=2.5)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--two-pass", action="store_true", help="separate ExtractChekpointRepr pass, as the reference loop")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:      # no launcher around us: start the ranks as children (bench.py)
        from bench import self_launch
        sys.exit(self_launch(sys.argv[1:], a.gpus, script=__file__))
    rank, local, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("VAMPIC_DIST_BACKEND", "nccl")
        ndev = max(torch.cuda.device_count(), 1)
        if backend == "nccl":
            torch.cuda.set_device(local % ndev)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local % ndev))
        else:
            dist.init_process_group(backend)
    assert world == a.gpus and torch.cuda.is_available()
    dev = torch.device("cuda", local % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(dev)
    from vampic import sharding
    forced = dist is None and sharding.init_single_rank_group(dev)     # VAMPIC_FORCE_COLLECTIVES=1: a 1-rank RCCL group
    if forced:
        import torch.distributed as dist
    rec = measure(dev, rank, world, dist, a.batch, a.size, a.steps, a.warmup, a.quality, a.no_graph, a.two_pass, forced)
    if rank == 0:
        print(json.dumps(rec), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
