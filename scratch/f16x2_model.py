"""Whole inference step in one arithmetic mode: stage tensors dumped per mode, then compared pairwise on the CPU.
    VAMPIC_CONV=f32   python scratch/f16x2_model.py dump m0      # fp32 matrix pipe
                      python scratch/f16x2_model.py dump m1      # bf16x3 (default)
    VAMPIC_CONV=f16x2 python scratch/f16x2_model.py dump m3      # fp16x2 (+ timing at 32x256x256)
                      python scratch/f16x2_model.py compare m0 m1 m3
VAMPIC_AMAX_CHECK=1 verifies every tracked max cell against a reduction of the tensor (slow, host syncs).
"""
import math, os, sys, time, torch
sys.path.insert(0, "/root/repo")
argv, sys.argv = sys.argv[1:], sys.argv[:1]
OUT = "/root/repo/gpurun_out/f16x2_%s.pt"

if argv[0] == "compare":
    d = {n: torch.load(OUT % n) for n in argv[1:]}
    names = argv[1:]
    for i in range(len(names)):
        for j in range(i + 1, len(names)):
            a, b = d[names[i]], d[names[j]]
            print(f"--- {names[i]} vs {names[j]}")
            for k in a:
                if k in ("bpp", "psnr"):
                    print(f"  {k}: {a[k]:.6f} vs {b[k]:.6f}  (d {b[k] - a[k]:+.2e})")
                    continue
                x, y = a[k], b[k]
                if x.dtype in (torch.int32, torch.uint8, torch.bool, torch.int64):
                    print(f"  {k}: {int((x != y).sum())} of {x.numel()} differ")
                else:
                    x, y = x.double(), y.double()
                    rms = float(x.pow(2).mean().sqrt())
                    print(f"  {k}: rms diff / rms {float((x - y).pow(2).mean().sqrt()) / rms:.3e}   max|d| / rms {float((x - y).abs().max()) / rms:.3e}")
    sys.exit(0)

import vampic
from vampic import ops, _lib as L
from bench import build_model

dev = torch.device("cuda")
net, sd = build_model(dev)
print("conv mode", L.load().vam_conv_get_mode(), flush=True)
with torch.no_grad():
    B, H, W, q = 4, 256, 256, 2.5
    x = vampic.synth.synth_image(B, H, W, 100 + B).to(dev)
    net.use_graph = False
    o = net.forward_single_quality(x, q)
    torch.cuda.synchronize()
    fp = list(net._plans.values())[-1]
    lik = o["likelihoods"]
    bpp = sum(float(torch.log2(v.double()).sum()) for v in lik.values()) / (-B * H * W)
    mse = float(((o["x_hat"].clamp(0, 1) - x) ** 2).mean())
    res = {"bpp": bpp, "psnr": -10 * math.log10(mse)}
    for name in ("y", "z_hat", "mu_b", "std_b", "mu_p", "std_p", "mask"):
        t = getattr(fp, name, None)
        if t is not None:
            res[name] = (t.buf if hasattr(t, "buf") else t).detach().cpu()
    res = {k: v for k, v in res.items() if v is not None}
    res["x_hat"] = o["x_hat"].cpu()
    res["lik_y"] = lik["y"].cpu()
    print({k: (round(v, 6) if isinstance(v, float) else tuple(v.shape)) for k, v in res.items()}, flush=True)
    torch.save(res, OUT % argv[1])
if os.environ.get("VAMPIC_AMAX_CHECK", "0") == "1":
    print("max-cell check passed", flush=True)
    sys.exit(0)
with torch.no_grad():
    x = vampic.synth.synth_image(32, 256, 256, 100).to(dev)
    for graph in (False, True):
        net.use_graph = graph
        for _ in range(3):
            net.forward_single_quality(x, 2.5, clone=False)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            net.forward_single_quality(x, 2.5, clone=False)
        torch.cuda.synchronize()
        print(f"32x256x256 q=2.5 graph={graph}: {(time.perf_counter() - t) * 100:.2f} ms/step", flush=True)
    plan = [p for p in net._plans.values()][-1].plan
    n_abs = sum(1 for m in plan.meta if m["desc"] == "absmax")
    print("plan steps", len(plan.steps), "absmax fallback steps", n_abs, "max cells used", getattr(plan, "_amax_next", 0), flush=True)
