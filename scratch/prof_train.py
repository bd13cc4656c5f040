import sys, argparse, collections
sys.path.insert(0, "/root/repo")
import torch, vampic
args = argparse.Namespace(model="rem", check_levels=[0.75], mu_std=True, dimension="middle", N=192, M=640,
                          multiple_decoder=True, multiple_encoder=True, multiple_hyperprior=True, dim_chunk=32,
                          division_dimension=[320, 640], mask_policy="point-based-std", support_progressive_slices=5,
                          delta_encode=True, total_mu_rep=True, all_scalable=True)
net = vampic.get_model(args, "cpu")
torch.nn.Module.load_state_dict(net, vampic.synth.synth_state_dict(net.state_dict(), seed=0))
net = net.cuda().train(); net.freeze_all(); net.unfreeze_rems(); net.use_graph = False
x = vampic.synth.synth_image(16, 256, 256, seed=1).cuda()
out = net.forward_finetune(x, 2.5)
plan = [p for k, p in net._plans.items() if "train" in k][0]
with torch.cuda.stream(plan.stream):
    prof = plan.plan.profile(3)
tot = sum(p["ms"] for p in prof)
print("total", tot, "steps", len(prof))
agg = collections.OrderedDict()
for p in prof:
    k = p["kind"] + " " + p["desc"][:60]
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += p["ms"]
for k, (n, ms) in sorted(agg.items(), key=lambda t: -t[1][1])[:40]:
    print(f"{ms:8.3f} ms  x{n:3d}  {k}")
with torch.cuda.stream(plan.stream):
    prof = plan.bwd.profile(3)
print("bwd total", sum(p["ms"] for p in prof), len(prof))
agg = collections.OrderedDict()
for p in prof:
    k = p["kind"] + " " + p["desc"][:60]
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += p["ms"]
for k, (n, ms) in sorted(agg.items(), key=lambda t: -t[1][1])[:12]:
    print(f"{ms:8.3f} ms  x{n:3d}  {k}")
