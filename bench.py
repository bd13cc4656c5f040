#!/usr/bin/env python3
"""Throughput of the codec hot path (BASELINE.json metric: megapixels/s encode+decode,
g_a -> hyperprior -> conditional entropy parameters -> variance mask -> likelihood -> LRP -> g_s)
on N MI355X GPUs of one node.

A step = one ``forward_single_quality(x, q=2.5)`` over a resident batch of 32 x 3 x 256 x 256
synthetic images (BASELINE.json configs[1]); weights are the deterministic synthetic set of
``vampic.synth`` (no checkpoints offline).  N > 1: one process per GPU, image batches shard
across ranks (independent units, no data-path collective): weak scaling, whole-job MP/s.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline      dominant kernel (conv_igemm, fp32 MFMA) measured live with HIP events on the
                launch stream: algorithmic FLOP / summed launch time vs the 157.3 TF fp32 matrix peak
  cpu_baseline  the CPU oracle (port of the reference's math, ATen CPU ops) on a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: bf16 dense (not the 2:1-sparsity figure)
BF16X3_PRODUCTS = 6                # bf16 partial products per fp32 multiply-add in the split-operand kernel
FLOP_PER_PIXEL = 1_784_853         # BASELINE.md §2, forward_single_quality 0 < q <= 10


def build_model(device):
    import vampic
    args = argparse.Namespace(model="pic", N=192, M=640, multiple_decoder=True, multiple_encoder=True,
                              multiple_hyperprior=True, dim_chunk=32, division_dimension=[320, 640],
                              mask_policy="point-based-std", support_progressive_slices=5, delta_encode=True,
                              total_mu_rep=True, all_scalable=True)
    net = vampic.get_model(args, "cpu").eval()
    sd = vampic.synth.synth_state_dict(net.state_dict(), seed=0)
    net.load_state_dict(sd)
    return net.to(device), sd


def cpu_baseline(sd, H, W, quality, budget_s=20.0):
    """Oracle (kind="port") on the host cores: as many 256x256 images as fit ~budget_s."""
    import vampic
    import vampic_oracle as O
    cores = int(os.environ.get("VAMPIC_CPU_THREADS", "16"))   # one-GPU box CPU share
    torch.set_num_threads(cores)
    x1 = vampic.synth.synth_image(1, H, W, seed=7)
    t0 = time.perf_counter()
    O.forward_single_quality(sd, x1, quality)                      # warm-up + calibration
    t1 = time.perf_counter() - t0
    nb = max(1, min(8, int(budget_s / 3 / max(t1, 1e-3))))
    x = vampic.synth.synth_image(nb, H, W, seed=8)
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        O.forward_single_quality(sd, x, quality)
        times.append(time.perf_counter() - t0)
    best = min(times)
    return {"value": nb * H * W / 1e6 / best, "unit": "MP/s", "cores": cores, "kind": "port",
            "sample": f"oracle forward_single_quality q={quality} on {nb}x3x{H}x{W}, best of 3, torch CPU fp32"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--quality", type=float, default=2.5)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL ("nccl") over xGMI on the 8-GPU node.  VAMPIC_DIST_BACKEND=gloo is for rehearsing the
        # multi-rank code path on a one-GPU box (all ranks then share cuda:0; RCCL refuses duplicate GPUs).
        backend = os.environ.get("VAMPIC_DIST_BACKEND", "nccl")
        ndev = max(torch.cuda.device_count(), 1)
        if backend == "nccl":
            torch.cuda.set_device(local % ndev)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local % ndev))
        else:
            dist.init_process_group(backend)
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU: the hot path has no CPU fallback"
    dev = torch.device("cuda", local % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(dev)
    red_dev = dev if (dist is None or dist.get_backend() == "nccl") else "cpu"

    import vampic
    from vampic import ops, _lib as L_
    net, sd = build_model(dev)
    net.use_graph = not a.no_graph
    B, H, W, q = a.batch, a.height, a.width, a.quality
    x = vampic.synth.synth_image(B, H, W, seed=100 + rank).to(dev)      # resident in HBM before timing

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    with torch.no_grad():
        for _ in range(max(a.warmup, 1)):
            out = net.forward_single_quality(x, q, clone=False)
        sync_all()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            out = net.forward_single_quality(x, q, clone=False)
        sync_all()
        dt = time.perf_counter() - t0
    from vampic import sharding
    dt = sharding.max_over_ranks(dt, red_dev)                 # slowest rank (RCCL all-reduce MAX)
    ms_step = dt / a.steps * 1e3
    mp_s = sharding.whole_job_megapixels_per_s(B, H, W, a.steps, world, dt)

    # ---- roofline of the dominant kernel, measured live with HIP events (eager replay, same plan)
    roof = None
    if rank == 0:
        net.use_graph = False
        ops.prof_reset()
        ops.prof_enable(True)
        psteps = 3
        with torch.no_grad():
            for _ in range(psteps):
                net.forward_single_quality(x, q, clone=False)
        torch.cuda.synchronize(dev)
        prof = ops.prof_read()
        ops.prof_enable(False)
        net.use_graph = not a.no_graph
        c = prof["conv_igemm"]
        achieved = c["flops"] / (c["ms"] * 1e-3) / 1e12 if c["ms"] > 0 else 0.0
        traffic = None
        if (B, H, W) == (32, 256, 256):
            # HBM-side bytes per launch from the committed rocprofv3 --pmc passes of this same command
            # (separate FETCH_SIZE / WRITE_SIZE runs, gfx950 x2 read correction): scripts/pmc_summary.py
            import glob
            pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
            if pm:
                t = json.load(open(pm[-1]))["conv_igemm"]
                traffic = round(t["traffic_bytes_per_step"] / max(t["launches_per_step"], 1))
        split = L_.load().vam_conv_get_mode() == 1
        # bf16x3 mode: every algorithmic (fp32-equivalent) FLOP costs 6 bf16 MFMA FLOPs, so the ceiling of this
        # algorithm on the bf16 pipe is 2500 / 6 = 416.7 TF/s; `achieved` stays ALGORITHMIC FLOP / time
        peak = BF16_MFMA_PEAK_TFLOPS / BF16X3_PRODUCTS if split else FP32_MFMA_PEAK_TFLOPS
        roof = {"bound": "mfma", "kernel": "conv_igemm_kernel (all launches of one step)",
                "arithmetic": ("fp32 operands split exactly into 3 bf16 terms, 6 partial products on v_mfma_f32_32x32x16_bf16, "
                               "fp32 accumulate") if split else "fp32 operands on v_mfma_f32_32x32x2_f32",
                "achieved": round(achieved, 3), "peak": round(peak, 1), "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4),
                "executed_mfma_tflops": round(achieved * (BF16X3_PRODUCTS if split else 1), 1),
                "vs_fp32_matrix_peak": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                "traffic": traffic, "traffic_unit": "bytes beyond L2 per launch (avg)",
                "launches_per_step": c["launches"] // psteps,
                "avg_launch_us": round(c["ms"] * 1e3 / max(c["launches"], 1), 2),
                "flop_per_step": c["flops"] / psteps,
                "algorithmic_bytes_per_launch": round(c["bytes"] / max(c["launches"], 1)),
                "kernel_ms_per_step": {k: round(v["ms"] / psteps, 3) for k, v in prof.items()},
                "whole_step_tflops": round(FLOP_PER_PIXEL * B * H * W / (ms_step * 1e-3) / 1e12, 3)}

    if rank == 0:
        bpp = -out["log2_likelihood_sum"].sum().item() / (B * H * W)
        line = {"metric": "megapixels/sec encode+decode (g_a->mask->g_s) at 256x256 bs32",
                "value": round(mp_s, 3), "unit": "MP/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32 (bf16x3 split operands)" if L_.load().vam_conv_get_mode() == 1 else "f32",
                "data": "synthetic",
                "config": {"workload": f"forward_single_quality q={q} on {B}x3x{H}x{W} per GPU "
                                       "(dual g_a, hyperprior, 10 base + 10 progressive slices, variance mask, "
                                       "likelihood, LRP, g_s[1]); README model N=192 M=640",
                           "batch_per_gpu": B, "global_batch": B * world, "quality": q,
                           "hip_graph": not a.no_graph, "weights": "synthetic seed 0", "bpp_check": round(bpp, 6)},
                "roofline": roof}
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sd, H, W, q)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
