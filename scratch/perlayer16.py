"""Per-launch time of the inference plan with the tile each conv launch chose (HIP events, launches serialised)."""
import sys, torch, collections, ctypes
sys.path.insert(0, "/root/repo")
import vampic
from vampic import _lib
sys.argv = sys.argv[:1]
from bench import build_model
net, sd = build_model(torch.device("cuda"))
lib = _lib.load()
x = vampic.synth.synth_image(32, 256, 256, 100).cuda()
with torch.no_grad():
    net.use_graph = False; net.storage = "bf16"
    net.forward_single_quality(x, 2.5, clone=False)
    plan = list(net._plans.values())[0].plan
    best = [float("inf")] * len(plan.steps)
    tiles = [None] * len(plan.steps)
    for _ in range(3):
        evs = []
        for i, s in enumerate(plan.steps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            if not isinstance(s, tuple):
                s()
                bm, bn, bk = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
                lib.vam_conv_last_tile(ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(bk))
                tiles[i] = (bm.value, bn.value)
            b.record()
            evs.append((a, b))
        torch.cuda.current_stream().synchronize()
        best = [min(t, a.elapsed_time(b)) for t, (a, b) in zip(best, evs)]
rows = [dict(m, ms=t, tile=tl, cls=c) for m, t, tl, c in zip(plan.meta, best, tiles, plan.class_of)]
tot = sum(r["ms"] for r in rows)
print(f"total {tot:.2f} ms over {len(rows)} steps; conv flops {sum(r['flops'] for r in rows)/1e12:.3f} T")
agg = collections.OrderedDict()
for r in rows:
    k = (r["desc"], r["tile"] if r["flops"] else None, r["cls"])
    a = agg.setdefault(k, [0, 0.0, 0.0]); a[0] += 1; a[1] += r["ms"]; a[2] += r["flops"]
print(f"{'desc':44s} {'tile':>9s} {'cls':>3s} {'n':>4s} {'ms':>8s} {'%':>6s} {'TF/s':>7s}")
for (k, tl, c), (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f"{k:44s} {str(tl):>9s} {c:3d} {n:4d} {ms:8.3f} {100*ms/tot:6.2f} {fl/ms/1e9 if ms else 0:7.1f}")
bt = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in rows:
    if r["flops"]:
        a = bt[r["tile"]]; a[0] += 1; a[1] += r["ms"]; a[2] += r["flops"]
for tl, (n, ms, fl) in sorted(bt.items(), key=lambda kv: -kv[1][1]):
    print(f"tile {tl}: {n} launches {ms:.3f} ms {fl/ms/1e9:.1f} TF/s")
