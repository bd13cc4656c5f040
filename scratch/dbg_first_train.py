"""Debug: GPU training forward vs the oracle's, tensor by tensor (run on the GPU box)."""
import argparse, os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import vampic, vampic.synth as synth, vampic_oracle as O
from conftest import README_ARGS
warnings.simplefilter("ignore")
net = vampic.get_model(argparse.Namespace(model="pic", **README_ARGS), "cpu")
sd = synth.synth_state_dict(net.state_dict(), seed=0)
net.load_state_dict(sd); net = net.cuda().train()
x = synth.synth_image(2, 64, 64, seed=5); ny = synth.uniform((2, 640, 4, 4), 201) - 0.5; nz = synth.uniform((2, 192, 1, 1), 202) - 0.5
mode = sys.argv[1] if len(sys.argv) > 1 else "multi"
if mode == "multi":
    out = net(x.cuda(), quality=[0, 10], training=True, noise={"y": ny, "z": nz})
    with torch.no_grad():
        ref = O.training_forward(sd, x, [0, 10], ny, nz)
else:
    out = net.forward_single_quality(x.cuda(), 2.5, training=True, noise={"y": ny, "z": nz})
    with torch.no_grad():
        ref = O.training_forward(sd, x, [2.5], ny, nz, single=True)
plan = [p for k, p in net._plans.items() if k[0] == "full_train"][0]
def cmp(name, a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    d = (a - b).abs()
    print(f"{name:12s} max|d| {float(d.max()):.3e}  rel {float(d.max() / (b.abs().max() + 1e-30)):.3e}  n(|d|>0.4) {int((d > 0.4).sum())}  n(|d|>1e-3) {int((d > 1e-3).sum())} of {d.numel()}")
    return d
cmp("y", plan.y.torch_nchw(), ref["y"])
cmp("z", plan.z.torch_nchw(), ref["z"])
cmp("mu_base", out["mu_base"], ref["mu_base"])
cmp("std_base", out["std_base"], ref["std_base"])
d = cmp("y_base", out["y_base"], ref["y_base"])
for i in range(10):
    di = d[:, 32 * i:32 * (i + 1)]
    print("   base slice", i, "max", float(di.max()), "n>0.4", int((di > 0.4).sum()))
if "y_prog" in ref:
    cmp("mu", plan.mu_p.torch_nchw(), ref["mu"]); cmp("std", plan.std_p.torch_nchw(), ref["std"])
    d = cmp("y_prog", out["y_prog"] if not isinstance(out["y_prog"], list) else out["y_prog"], ref["y_prog"])
    if "mask" in ref:
        print("mask xor", int((plan.mask.torch_nchw().cpu() != ref["mask"]).sum()))
lik = out["likelihoods"]
cmp("lik_y", lik["y"], ref["likelihoods"]["y"])
if "y_prog" in lik: cmp("lik_y_prog", lik["y_prog"], ref["likelihoods"]["y_prog"])
cmp("x_hat", out["x_hat"], ref["x_hat"])
# where do the roundings differ, and are those boundary events in the oracle's own numbers?
yb_g, yb_r = out["y_base"].cpu(), ref["y_base"]
r = (ref["y"][:, :320] - ref["mu_base"])
frac = (r - torch.floor(r) - 0.5).abs()
flips = (yb_g - yb_r).abs() > 0.4
print("base flips", int(flips.sum()), "their distance to x.5 in the oracle:", frac[flips][:10].tolist())
