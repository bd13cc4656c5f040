#!/usr/bin/env python3
"""Throughput of the codec hot path (BASELINE.json metric: megapixels/s encode+decode,
g_a -> hyperprior -> conditional entropy parameters -> variance mask -> likelihood -> LRP -> g_s)
on N MI355X GPUs of one node.

A step = one ``forward_single_quality(x, q=2.5)`` over a resident batch of 32 x 3 x 256 x 256
synthetic images (BASELINE.json configs[1]); weights are the deterministic synthetic set of
``vampic.synth`` (no checkpoints offline).  N > 1: one process per GPU, image batches shard
across ranks (independent units, no data-path collective): weak scaling, whole-job MP/s.

``python bench.py --gpus N`` without a launcher starts its own N ranks (children of this process, through
``python -m torch.distributed.run``, before anything here touches the GPU) and relays rank 0's line; under an
external ``torch.distributed.run`` (RANK / WORLD_SIZE set) it is one of the ranks.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline      dominant kernel (conv_igemm_kernel) measured live with HIP events on the launch stream: algorithmic
                FLOP / summed launch time.  Default arithmetic: every fp32 operand split exactly into three bf16 terms,
                six partial products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation -> ceiling 2500 / 6 = 416.7
                TF/s fp32-equivalent (VAMPIC_CONV=f32: fp32 operands on v_mfma_f32_32x32x2_f32, peak 157.3 TF/s).
                ``roofline.g_a_g_s`` is the same quotient over the g_a / g_s launches only (the stack the north star's
                40 % target names) and ``roofline.classes`` the per-class split (ms, TF/s per step).
  cpu_baseline  the CPU oracle (port of the reference's math, ATen CPU ops) on a bounded sample, all host cores this
                process may use, median of 5 runs after 2 warm-ups
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
POWER_LIMITED_SPLIT_TFLOPS = 266.8  # profiles/r03_mfma_shapes.txt: MFMA-only inner loop on Gaussian operands, fp32-equivalent
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


def host_cores() -> int:
    """Cores this process may actually use: the smaller of its affinity mask and its cgroup CPU quota (the one-GPU box
    hands out a CPU share of a larger host; asking ATen for more threads than the quota only thrashes)."""
    if os.environ.get("VAMPIC_CPU_THREADS"):
        return int(os.environ["VAMPIC_CPU_THREADS"])
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:                                                   # cgroup v2: "<quota> <period>" or "max <period>"
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:                                               # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sd, H, W, quality, budget_s=24.0):
    """Oracle (kind="port") on the host cores: as many images as keep 2 warm-ups + 5 timed runs inside
    ~budget_s; the median run is reported."""
    import statistics
    import vampic
    import vampic_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline: oracle on {cores} host threads ({cpu_model()})", file=sys.stderr, flush=True)
    x1 = vampic.synth.synth_image(1, H, W, seed=7)
    O.forward_single_quality(sd, x1, quality)                      # first call: page-in, thread pool start
    t0 = time.perf_counter()
    O.forward_single_quality(sd, x1, quality)                      # calibration
    t1 = time.perf_counter() - t0
    nb = max(1, min(8, int(budget_s / 7 / max(t1, 1e-3))))
    print(f"[bench] cpu_baseline: {t1:.2f} s per image -> sample of {nb} images x 7 runs", file=sys.stderr, flush=True)
    x = vampic.synth.synth_image(nb, H, W, seed=8)
    for _ in range(2):
        O.forward_single_quality(sd, x, quality)
    times = []
    for _ in range(5):
        t0 = time.perf_counter()
        O.forward_single_quality(sd, x, quality)
        times.append(time.perf_counter() - t0)
    med = statistics.median(times)
    return {"value": nb * H * W / 1e6 / med, "unit": "MP/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "sample": f"oracle forward_single_quality q={quality} on {nb}x3x{H}x{W}, median of 5 after 2 warm-ups, "
                      f"torch CPU fp32, {cores} threads"}


def self_launch(argv, n: int, script: str = None) -> int:
    """``python bench.py --gpus N`` with no launcher around it: start N ranks as CHILD processes (one per GPU)
    through torch.distributed.run — nothing in this parent has touched the GPU — relay their output and
    return their exit code.  Rank 0 prints the one JSON line."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["VAMPIC_BENCH_CHILD"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(script or __file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def train_legs(dev, dist, forced):
    """BASELINE configs[3] / [4] on one GPU's share: a first_train step (forward [0, 10] of 32x3x256x256 + backward of all
    150 M parameters + clip + Adam; scripts/bench_train.py) and a REM fine-tune step (16x3x256x256, q = 2.5, check level
    0.75; scripts/bench_finetune.py), timed by those scripts' own ``measure`` (hipGraph replay, 2 warm-up + 5 timed
    steps).  Returns {"first_train": {...}, "rem_finetune": {...}} for the line's ``train`` object."""
    import gc
    import importlib.util

    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "scripts", name + ".py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m

    out = {}
    for key, mod, kw in (("first_train", "bench_train", dict(batch=32, size=256, steps=5, warmup=2)),
                         ("rem_finetune", "bench_finetune", dict(batch=16, size=256, steps=5, warmup=2))):
        t0 = time.perf_counter()
        print(f"[bench] train leg {key} ...", file=sys.stderr, flush=True)
        r = load(mod).measure(dev, 0, 1, dist, forced=forced, **kw)
        rec = {"workload": r["config"]["workload"], "ms_per_step": r["ms_per_step"],
               "ms_per_step_gc_unfrozen": r["ms_per_step_gc_unfrozen"], "images_per_s": r["value"], "steps": r["steps"],
               "warmup": r["warmup"], "phase_ms": r["phase_ms"], "dtype": r["dtype"], "collectives": r["collectives"],
               "loss": r["config"]["loss"], "trainable_params": r["config"]["trainable_params"]}
        if "algorithmic_tflops" in r:
            rec["tflops"] = r["algorithmic_tflops"]
            rec["frac_of_split_ceiling"] = round(r["algorithmic_tflops"] / (BF16_MFMA_PEAK_TFLOPS / BF16X3_PRODUCTS), 4)
        out[key] = rec
        gc.collect()
        print(f"[bench] train leg {key}: {rec['ms_per_step']} ms/step ({time.perf_counter() - t0:.0f} s incl. plan build)",
              file=sys.stderr, flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--quality", type=float, default=2.5)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="f32 (default, the headline and every parity claim) or bf16: BASELINE configs[2] storage mode — g_a / g_s "
                         "feature maps >= 64x64 stored in bf16, bf16-rounded weights there, fp32 accumulation; its distance to "
                         "the fp32 path (mask XOR, dPSNR, dbpp) is measured and printed in the line")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bf16", action="store_true", help="skip the secondary bf16-storage measurement of the default run")
    ap.add_argument("--no-train", action="store_true",
                    help="skip the secondary training measurements of the default run (BASELINE configs[3] / [4]: first_train and "
                         "REM fine-tune steps on this GPU's share of the batch)")
    ap.add_argument("--dry", action="store_true",
                    help="rehearse rank setup / sharding / aggregation without touching a GPU (CPU test of the N>1 entry)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(sys.argv[1:], a.gpus))

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
    if a.dry:
        return dry_run(a, rank, world, dist)
    assert torch.cuda.is_available(), "bench.py needs a GPU: the hot path has no CPU fallback"
    dev = torch.device("cuda", local % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(dev)
    from vampic import sharding
    forced = dist is None and sharding.init_single_rank_group(dev)     # VAMPIC_FORCE_COLLECTIVES=1: a 1-rank RCCL group
    if forced:
        import torch.distributed as dist
    red_dev = dev if (dist is None or dist.get_backend() == "nccl") else "cpu"

    import vampic
    from vampic import ops, _lib as L_
    net, sd = build_model(dev)
    net.use_graph = not a.no_graph
    net32 = None
    if a.dtype == "bf16":
        import copy
        net32 = net
        net = copy.deepcopy(net)
        net.storage = "bf16"
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
        classes = ops.prof_read_classes()
        ops.prof_enable(False)
        net.use_graph = not a.no_graph
        c = prof["conv_igemm"]
        achieved = c["flops"] / (c["ms"] * 1e-3) / 1e12 if c["ms"] > 0 else 0.0
        traffic, traffic_src = None, None
        if (B, H, W) == (32, 256, 256):
            # NOT measured in this run: PMC counters need rocprofv3 around the process.  The figure is read from the
            # committed summary of the rocprofv3 --pmc passes of this same command (separate FETCH_SIZE / WRITE_SIZE
            # runs, gfx950 x2 read correction: scripts/pmc_summary.py); `traffic_source` names the file.
            import glob
            pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
            if pm:
                t = json.load(open(pm[-1]))["conv_igemm"]
                traffic = round(t["traffic_bytes_per_step"] / max(t["launches_per_step"], 1))
                traffic_src = ("committed rocprofv3 --pmc summary " + os.path.relpath(pm[-1], ROOT) +
                               " (separate FETCH_SIZE/WRITE_SIZE passes of this command; not measured by this run)")
        split = L_.load().vam_conv_get_mode() == 1
        # bf16x3 mode: every algorithmic (fp32-equivalent) FLOP costs 6 bf16 MFMA FLOPs, so the ceiling of this
        # algorithm on the bf16 pipe is 2500 / 6 = 416.7 TF/s; `achieved` stays ALGORITHMIC FLOP / time
        peak = BF16_MFMA_PEAK_TFLOPS / BF16X3_PRODUCTS if split else FP32_MFMA_PEAK_TFLOPS
        roof = {"bound": "mfma", "kernel": "conv_igemm_kernel + resunit192_kernel (all convolution launches of one step; a fused residual unit is one launch of three convolutions)",
                "arithmetic": ("fp32 operands split exactly into 3 bf16 terms, 6 partial products on v_mfma_f32_32x32x16_bf16, "
                               "fp32 accumulate") if split else "fp32 operands on v_mfma_f32_32x32x2_f32",
                "achieved": round(achieved, 3), "peak": round(peak, 1), "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4),
                "executed_mfma_tflops": round(achieved * (BF16X3_PRODUCTS if split else 1), 1),
                "vs_fp32_matrix_peak": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                "traffic": traffic, "traffic_unit": "bytes beyond L2 per launch (avg)", "traffic_source": traffic_src,
                "launches_per_step": c["launches"] // psteps,
                "avg_launch_us": round(c["ms"] * 1e3 / max(c["launches"], 1), 2),
                "flop_per_step": c["flops"] / psteps,
                "algorithmic_bytes_per_launch": round(c["bytes"] / max(c["launches"], 1)),
                "kernel_ms_per_step": {k: round(v["ms"] / psteps, 3) for k, v in prof.items()},
                "whole_step_tflops": round(FLOP_PER_PIXEL * B * H * W / (ms_step * 1e-3) / 1e12, 3)}
        # per-class split of the conv launches (set by the plan: which part of the path a launch belongs to)
        cls = {}
        for k, v in classes.items():
            if v["launches"]:
                cls[k] = {"ms_per_step": round(v["ms"] / psteps, 3), "launches_per_step": v["launches"] // psteps,
                          "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["ms"] > 0 else 0.0}
        roof["classes"] = cls
        if split:
            # NOT measured by this run: the matrix pipe's rate on Gaussian operands, where the clock is power-limited.
            # scratch/probe/mfma_shapes.hip runs the consumer wave's inner loop alone (LDS fragment reads + the six
            # products of a 128 x 192 tile, no global traffic) on all 256 CUs: 1,600 TF/s executed with Gaussian planes,
            # 2,360 TF/s with all-zero planes (profiles/r03_mfma_shapes.txt).  `frac` above stays against the nominal peak.
            roof["power_limited_ceiling"] = {
                "tflops": POWER_LIMITED_SPLIT_TFLOPS, "frac": round(achieved / POWER_LIMITED_SPLIT_TFLOPS, 4),
                "source": "committed probe log profiles/r03_mfma_shapes.txt (1,600 TF/s executed bf16 MFMA / 6 products; "
                          "not measured by this run)"}
        gms = classes["g_a"]["ms"] + classes["g_s"]["ms"]
        gfl = classes["g_a"]["flops"] + classes["g_s"]["flops"]
        if gms > 0:
            ga = gfl / (gms * 1e-3) / 1e12
            roof["g_a_g_s"] = {"what": "all convolution launches of g_a[0], g_a[1] and g_s[1] (the stack the north star's "
                                       ">= 40 % target names), same HIP-event measurement",
                               "achieved": round(ga, 3), "peak": round(peak, 1), "unit": "TFLOP/s",
                               "frac": round(ga / peak, 4), "ms_per_step": round(gms / psteps, 3),
                               "flop_per_step": gfl / psteps}

    vs_fp32 = None
    if rank == 0 and net32 is not None:
        with torch.no_grad():
            o32 = net32.forward_single_quality(x, q, clone=True)
            o16 = net.forward_single_quality(x, q, clone=True)
        ps = lambda o: -10.0 * torch.log10(torch.mean((x - o["x_hat"]) ** 2)).item()
        bp = lambda o: -o["log2_likelihood_sum"].sum().item() / (B * H * W)
        vs_fp32 = {"reference": "the fp32 HIP path on the same weights and inputs (itself bit-exact in mask / symbols against the "
                                "CPU oracle: tests/test_gpu_model.py)",
                   "mask_xor": int((o16["mask"] != o32["mask"]).sum()), "mask_elements": o32["mask"].numel(),
                   "d_psnr_db": round(ps(o16) - ps(o32), 5), "d_bpp": round(bp(o16) - bp(o32), 6),
                   "psnr_fp32_db": round(ps(o32), 4), "bpp_fp32": round(bp(o32), 5)}
    # ---- secondary: the SAME steps under the bf16-storage configuration (BASELINE configs[2]'s arithmetic), timed the same
    # way on rank 0 of a 1-GPU run and compared with the fp32 result just measured.  The headline stays fp32.
    bf16_rec = None
    if rank == 0 and world == 1 and a.dtype == "f32" and not a.no_bf16:
        import copy
        net16 = copy.deepcopy(net)
        net16.storage = "bf16"
        with torch.no_grad():
            o32 = net.forward_single_quality(x, q, clone=True)
            for _ in range(max(a.warmup, 1)):
                o16 = net16.forward_single_quality(x, q, clone=False)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(a.steps):
                o16 = net16.forward_single_quality(x, q, clone=False)
            torch.cuda.synchronize(dev)
            dt16 = time.perf_counter() - t0
            o16 = net16.forward_single_quality(x, q, clone=True)
        ps = lambda o: -10.0 * torch.log10(torch.mean((x - o["x_hat"]) ** 2)).item()
        bp = lambda o: -o["log2_likelihood_sum"].sum().item() / (B * H * W)
        bf16_rec = {"what": "same workload with model.storage = 'bf16' (g_a / g_s feature maps >= 64x64 and the weights that touch "
                            "them stored in bf16, fp32 accumulation; entropy-parameter stacks, variance mask and likelihoods fp32); "
                            "differences are against this run's fp32 result on the same input — a measurement, not a parity claim",
                    "ms_per_step": round(dt16 / a.steps * 1e3, 3), "value": round(B * H * W * a.steps / 1e6 / dt16, 3), "unit": "MP/s",
                    "mask_xor": int((o16["mask"] != o32["mask"]).sum()), "mask_elements": o32["mask"].numel(),
                    "d_psnr_db": round(ps(o16) - ps(o32), 5), "d_bpp": round(bp(o16) - bp(o32), 6)}
        del net16
    # ---- secondary: the training configurations (BASELINE configs[3] first_train, configs[4] REM fine-tune) on this GPU's share
    # of their batches, 5 timed steps each, so that the driver's one line carries them too.  The headline is unchanged.
    train_rec = None
    if rank == 0 and world == 1 and a.dtype == "f32" and not a.no_train and (B, H, W) == (32, 256, 256):
        train_rec = train_legs(dev, dist, forced)
    if rank == 0:
        bpp = -out["log2_likelihood_sum"].sum().item() / (B * H * W)
        line = {"metric": "megapixels/sec encode+decode (g_a->mask->g_s) at 256x256 bs32",
                "value": round(mp_s, 3), "unit": "MP/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": ("bf16 storage (g_a/g_s feature maps >= 64x64 and the weights that touch them in bf16, fp32 accumulate; "
                          "entropy-parameter stacks, variance mask, likelihoods fp32)") if a.dtype == "bf16" else
                         ("f32 (bf16x3 split operands)" if L_.load().vam_conv_get_mode() == 1 else "f32"),
                "data": "synthetic",
                "config": {"workload": f"forward_single_quality q={q} on {B}x3x{H}x{W} per GPU "
                                       "(dual g_a, hyperprior, 10 base + 10 progressive slices, variance mask, "
                                       "likelihood, LRP, g_s[1]); README model N=192 M=640",
                           "batch_per_gpu": B, "global_batch": B * world, "quality": q,
                           "hip_graph": not a.no_graph, "weights": "synthetic seed 0", "bpp_check": round(bpp, 6),
                           "bpp_route": "in-kernel float64 sum of log2(likelihood) (tests hold it to |dbpp| <= 1e-6 ABSOLUTE against "
                                        "the float64 sum over the oracle's likelihoods on difference-free cases)"},
                "roofline": roof}
        if vs_fp32 is not None:
            line["vs_fp32"] = vs_fp32
        if bf16_rec is not None:
            line["bf16"] = bf16_rec
        if train_rec is not None:
            line["train"] = train_rec
        line["retired_graphs"] = ops.retired_graphs()      # executable graphs kept alive instead of destroyed (ops.Graph docstring)
        if forced:
            line["collectives"] = "forced (1-rank nccl group: barrier + max-over-ranks all-reduce issued on RCCL)"
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sd, H, W, q)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def dry_run(a, rank, world, dist):
    """N > 1 entry without a GPU: the ranks shard a job of images, "run" timed steps of a fixed host-side duration
    (rank r takes (1 + r) ms per step, so the aggregate must show the slowest rank), and rank 0 prints the line the
    driver parses.  Exercises exactly the launch / rendezvous / max-over-ranks / whole-job-rate code of the real run."""
    from vampic import sharding
    B, H, W = a.batch, a.height, a.width

    def sync_all():
        if dist is not None:
            dist.barrier()

    sync_all()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        time.sleep(1e-3 * (1 + rank))
    sync_all()
    dt = sharding.max_over_ranks(time.perf_counter() - t0, "cpu")
    b, e = sharding.shard_range(B * world, rank, world)
    n_img = sharding.sum_over_ranks([float(e - b)], "cpu")[0]
    if rank == 0:
        print(json.dumps({"metric": "megapixels/sec encode+decode (g_a->mask->g_s) at 256x256 bs32",
                          "value": round(sharding.whole_job_megapixels_per_s(B, H, W, a.steps, world, dt), 3),
                          "unit": "MP/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "none (dry run)", "data": "none", "dry": True,
                          "config": {"workload": "dry run: rank setup + aggregation only", "global_batch": int(n_img)}}),
              flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
