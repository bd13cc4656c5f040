"""Fused residual unit (csrc/resunit.hip) vs the three-launch path: us per group of K units at 32 x 64 x 64 x 192."""
import os, sys, torch
sys.path.insert(0, "/root/repo")
import vampic
from vampic import layers as Ly, ops, engine, _lib as L
lib = L.load()
K = int(os.environ.get("K", "4"))
mods = [Ly.ResidualUnit(192) for _ in range(K)]
for i, m in enumerate(mods):
    m.load_state_dict(vampic.synth.synth_state_dict(m.state_dict(), 40 + i))
    m.cuda()
xs = [ops.from_nchw(vampic.synth.normal((32, 192, 64, 64), 50 + i).cuda()) for i in range(K)]


def build(fused):
    os.environ["VAMPIC_FUSED_RU"] = "1" if fused else "0"
    plan = engine.Plan("cuda")
    outs = engine.lower_residual_units(plan, mods, xs)
    return plan, outs


def timeit(plan, reps=20):
    for _ in range(3):
        plan.run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        plan.run()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


p3, o3 = build(False)
pf, of = build(True)
flops = K * 2.0 * 32 * 64 * 64 * (192 * 96 * 2 + 96 * 96 * 9)
res = {}
for rnd in range(3):
    for name, plan, dma in (("three launches", p3, 1), ("fused, LDS-DMA ring", pf, 1), ("fused, register-staged", pf, 0)):
        lib.vam_resunit_set_dma(dma)
        res.setdefault(name, []).append(timeit(plan))
lib.vam_resunit_set_dma(-1)
pf.run(); p3.run(); torch.cuda.synchronize()
same = all(torch.equal(a.buf, b.buf) for a, b in zip(of, o3))
for name, ts in res.items():
    t = min(ts)
    print(f"{name:26s} {t:8.1f} us per {K} units  ({flops / t / 1e6:6.1f} TF/s)   rounds: {' '.join(f'{x:.0f}' for x in ts)}")
print("bit-identical:", same)
