#!/usr/bin/env python3
"""Secondary benchmark: first-stage training step (BASELINE.json configs[3]: `train.py first_train`, synthetic 256x256,
batch 256 sharded over 8 MI355X = 32 images per GPU, RCCL gradient all-reduce over xGMI).

A step = what training/step.py:56-99 does per batch with ``sampling_training=False``: zero_grad, ``model(d, quality=[0, 10])``
(training forward, both decoders), ScalableRateDistortionLoss, backward of EVERY parameter (150 M, HIP kernels), the
bucketed gradient all-reduce issued DURING the backward (~25 MB buckets in the order the backward finishes them; clip after
the last bucket), clip_grad_norm_ 1.0, Adam.  One process per GPU; launched like bench.py:

    python scripts/bench_train.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        scripts/bench_train.py --gpus N ...

Prints one JSON line: whole-job images/s (weak scaling), the per-phase split and the algorithmic TFLOP/s of the step
(138.14 GFLOP forward per 256x256 image at [0, 10], x3 with the backward: SURVEY 8d).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

FWD_GFLOP_PER_IMAGE_256 = 138.14          # SURVEY 8d, forward([0, 10]) on one 256x256 image


def measure(dev, rank=0, world=1, dist=None, batch=32, size=256, steps=5, warmup=2, no_graph=False, foreach_adam=False,
            forced=False, freeze_gc=True):
    """Time ``steps`` first_train steps on this rank's synthetic shard; returns the record rank 0 prints (bench.py embeds
    it in its own line as ``train.first_train``).  ``freeze_gc``: see the comment at gc.freeze() below — the un-frozen
    time is reported beside it as ``ms_per_step_gc_unfrozen`` (measured first, same steps)."""
    import vampic
    from vampic import finetune as ft, sharding
    from vampic.checkpoint import configure_optimizers
    args = argparse.Namespace(model="pic", N=192, M=640, multiple_decoder=True, multiple_encoder=True,
                              multiple_hyperprior=True, dim_chunk=32, division_dimension=[320, 640],
                              mask_policy="point-based-std", support_progressive_slices=5, delta_encode=True,
                              total_mu_rep=True, all_scalable=True, learning_rate=1e-4, aux_learning_rate=1e-3,
                              training_type="first_train")
    net = vampic.get_model(args, "cpu")
    torch.nn.Module.load_state_dict(net, vampic.synth.synth_state_dict(net.state_dict(), seed=0))
    net = net.to(dev).train()
    ft.first_train_setup(net)
    net.use_graph = not no_graph
    args.fused_adam = not foreach_adam
    opt, _ = configure_optimizers(net, args)
    crit = ft.ScalableRateDistortionLoss(lmbda_list=[0.0055, 0.04], device=dev)
    x = vampic.synth.synth_image(batch, size, size, seed=300 + rank).to(dev)

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(max(warmup, 1)):
        c = ft.first_train_step(net, crit, x, opt, [0, 10])
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        c = ft.first_train_step(net, crit, x, opt, [0, 10])
    sync()
    dt_unfrozen = time.perf_counter() - t0
    # The plans hold ~1e6 long-lived Python objects (problem structs, views): a full (generation-2) pass of the cyclic
    # collector over them takes ~150 ms and one can fall inside a short window (measured: scratch/ft_steps.py).  The
    # headline is measured with what exists now parked in the permanent generation, as long-running training loops do;
    # the same steps WITHOUT the freeze were timed just above and are reported beside it.
    import gc
    if freeze_gc:
        gc.collect()
        gc.freeze()
    t0 = time.perf_counter()
    for _ in range(steps):
        c = ft.first_train_step(net, crit, x, opt, [0, 10])
    sync()
    dt = sharding.max_over_ranks(time.perf_counter() - t0, dev if (dist is None or dist.get_backend() == "nccl") else "cpu")

    def timed(fn):
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        r = fn()
        torch.cuda.synchronize(dev)
        return r, (time.perf_counter() - t) * 1e3
    opt.zero_grad()
    out, t_fwd = timed(lambda: net(x, quality=[0, 10], training=True))
    loss = crit(out, x)["loss"]
    _, t_bwd = timed(loss.backward)                         # includes the bucketed all-reduce when world > 1
    _, t_opt = timed(lambda: (ft.clip_grad_norm_(net, 1.0), opt.step()))      # the step's own clip (one reduction over the flat buffer)
    plan = next(p for k, p in net._plans.items() if k[0] == "full_train")
    n_par = sum(p.numel() for p in net.parameters() if p.requires_grad)
    gc.unfreeze()
    gflop = 3.0 * FWD_GFLOP_PER_IMAGE_256 * (size * size / 65536.0) * batch
    return ({"metric": "first_train images/sec (256x256 patches, forward [0,10] + backward of 150 M parameters + Adam)",
                          "value": round(world * batch * steps / dt, 2), "unit": "images/s", "n_gpus": world,
                          "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 3), "ms_per_step_gc_unfrozen": round(dt_unfrozen / steps * 1e3, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 (bf16x3 split operands, forward, data and weight gradients)",
                          "data": "synthetic", "collectives": ("forced (1-rank nccl group)" if forced else ("nccl" if world > 1 else "none")),
                          "config": {"workload": f"first_train step, quality [0, 10], {batch}x3x{size}x{size} per GPU",
                                     "global_batch": batch * world, "trainable_params": n_par, "grad_bytes": 4 * n_par,
                                     "grad_buckets": len(plan.bucket_bounds), "hip_graph": not no_graph, "adam": "torch fused" if args.fused_adam else "torch foreach",
                                     "loss": round(float(c["loss"].detach()), 5)},
                          "algorithmic_tflops": round(gflop / (dt / steps) / 1e3, 2),
                          "phase_ms": {"train_forward": round(t_fwd, 3), "backward_incl_all_reduce": round(t_bwd, 3),
                                       "clip_adam": round(t_opt, 3)}})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--foreach-adam", action="store_true", help="torch.optim.Adam as the reference constructs it (default here: fused=True)")
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
    rec = measure(dev, rank, world, dist, a.batch, a.size, a.steps, a.warmup, a.no_graph, a.foreach_adam, forced)
    if rank == 0:
        print(json.dumps(rec), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
