"""Per-launch time of the first-stage training plans (forward and backward), aggregated by launch description."""
import sys, argparse, collections, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vampic
from vampic import finetune as ft
args = argparse.Namespace(model="pic", N=192, M=640, multiple_decoder=True, multiple_encoder=True, multiple_hyperprior=True, dim_chunk=32,
                          division_dimension=[320, 640], mask_policy="point-based-std", support_progressive_slices=5,
                          delta_encode=True, total_mu_rep=True, all_scalable=True)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
net = vampic.get_model(args, "cpu")
torch.nn.Module.load_state_dict(net, vampic.synth.synth_state_dict(net.state_dict(), seed=0))
net = net.cuda().train(); net.use_graph = False
x = vampic.synth.synth_image(B, 256, 256, seed=1).cuda()
crit = ft.ScalableRateDistortionLoss(lmbda_list=[0.0055, 0.04], device="cuda")
out = net(x, quality=[0, 10], training=True)
crit(out, x)["loss"].backward()
plan = [p for k, p in net._plans.items() if k[0] == "full_train"][0]
for name, pl in (("forward", plan.plan), ("backward", plan.bwd)):
    with torch.cuda.stream(plan.stream):
        prof = pl.profile(2)
    print(name, "total", round(sum(p["ms"] for p in prof), 2), "ms, steps", len(prof))
    agg = collections.OrderedDict()
    for p in prof:
        k = p["kind"] + " " + p["desc"][:70]
        a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += p["ms"]
    for k, (n, ms) in sorted(agg.items(), key=lambda t: -t[1][1])[:45]:
        print(f"{ms:9.3f} ms  x{n:3d}  {k}")
    kinds = collections.Counter()
    for p in prof:
        kinds[p["kind"]] += p["ms"]
    print({k: round(v, 2) for k, v in kinds.items()})
