import sys, torch, argparse, collections
sys.path.insert(0, "/root/repo")
import vampic
sys.argv = sys.argv[:1]
from bench import build_model
net, sd = build_model(torch.device("cuda"))
x = vampic.synth.synth_image(32, 256, 256, 100).cuda()
with torch.no_grad():
    net.use_graph = False
    net.forward_single_quality(x, 2.5, clone=False)
    plan = list(net._plans.values())[0]
    s = plan.stream
    with torch.cuda.stream(s):
        rows = plan.plan.profile(3)
tot = sum(r["ms"] for r in rows)
print(f"total {tot:.2f} ms over {len(rows)} steps; conv flops {sum(r['flops'] for r in rows)/1e12:.3f} T")
agg = collections.OrderedDict()
for r in rows:
    a = agg.setdefault(r["desc"], [0, 0.0, 0.0]); a[0] += 1; a[1] += r["ms"]; a[2] += r["flops"]
print(f"{'desc':48s} {'n':>4s} {'ms':>8s} {'%':>6s} {'TF/s':>7s}")
for k, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{k:48s} {n:4d} {ms:8.3f} {100*ms/tot:6.2f} {fl/ms/1e9 if ms else 0:7.1f}")
