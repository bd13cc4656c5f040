#!/usr/bin/env python3
"""Where a REM fine-tune step's wall time goes: per-step wall (sync after each), unsynchronised loop, host time per
call of the step, and a cProfile of 20 unsynchronised steps.  usage: python scratch/ft_steps.py [--no-graph]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-graph", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
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
    net.use_graph = not a.no_graph
    opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=1e-4)
    crit = ft.RateLoss()
    x = vampic.synth.synth_image(16, 256, 256, seed=200).to(dev)
    import gc
    gc.callbacks.append(lambda ph, info: print("   gc", ph, info, flush=True) if info.get("generation") == 2 else None)
    params = [p for p in net.parameters() if p.requires_grad]
    for i in range(14):
        ts = [time.perf_counter()]
        opt.zero_grad(); ts.append(time.perf_counter())
        out = net.forward_finetune(x, 2.5); ts.append(time.perf_counter())
        c = crit(out, x); ts.append(time.perf_counter())
        c["loss"].backward(); ts.append(time.perf_counter())
        torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0); ts.append(time.perf_counter())
        opt.step(); ts.append(time.perf_counter())
        torch.cuda.synchronize(); ts.append(time.perf_counter())
        print(f"step {i}:", " ".join(f"{n} {1e3 * (ts[j + 1] - ts[j]):.2f}" for j, n in
                                    enumerate(["zero", "fwd", "crit", "bwd", "clip", "adam", "sync"])), flush=True)
    # 2. unsynchronised loop
    t = time.perf_counter()
    for _ in range(20):
        ft.finetune_step(net, crit, x, opt, 2.5, [0.75])
    torch.cuda.synchronize()
    print(f"loop of 20: {1e3 * (time.perf_counter() - t) / 20:.2f} ms/step", flush=True)
    # 3. host time per call, no sync between
    acc = {}

    def T(name, fn):
        t = time.perf_counter()
        r = fn()
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
        return r
    params = [p for p in net.parameters() if p.requires_grad]
    for _ in range(10):
        T("zero_grad", opt.zero_grad)
        out = T("forward", lambda: net.forward_finetune(x, 2.5))
        c = T("criterion", lambda: crit(out, x))
        T("backward", c["loss"].backward)
        T("all_reduce", lambda: sharding.all_reduce_gradients(params))
        T("clip", lambda: torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0))
        T("adam", opt.step)
    torch.cuda.synchronize()
    print("host ms per call:", {k: round(v * 100, 3) for k, v in acc.items()}, flush=True)
    # 4. GPU time per phase with events (no host sync inside the step)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in range(10)]
    for i in range(10):
        opt.zero_grad()
        ev[i][0].record()
        out = net.forward_finetune(x, 2.5)
        ev[i][1].record()
        c = crit(out, x)
        ev[i][2].record()
        c["loss"].backward()
        ev[i][3].record()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
        ev[i][4].record()
        opt.step()
        ev[i][5].record()
    torch.cuda.synchronize()
    names = ["forward", "criterion", "backward", "clip", "adam"]
    print("gpu-timeline ms:", {n: round(sum(e[j].elapsed_time(e[j + 1]) for e in ev) / 10, 3) for j, n in enumerate(names)},
          "step-to-step", round(ev[0][0].elapsed_time(ev[9][0]) / 9, 3), flush=True)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        ft.finetune_step(net, crit, x, opt, 2.5, [0.75])
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()
