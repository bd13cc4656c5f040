"""Where a workgroup of the fused residual unit spends its cycles: s_memtime stamps at the phase boundaries."""
import os, sys, torch
sys.path.insert(0, "/root/repo")
import vampic
from vampic import layers as Ly, ops, engine, _lib as L
lib = L.load()
K = int(os.environ.get("K", "4"))
mods = [Ly.ResidualUnit(192) for _ in range(K)]
for i, m in enumerate(mods):
    m.load_state_dict(vampic.synth.synth_state_dict(m.state_dict(), 40 + i)); m.cuda()
xs = [ops.from_nchw(vampic.synth.normal((32, 192, 64, 64), 50 + i).cuda()) for i in range(K)]
plan = engine.Plan("cuda")
engine.lower_residual_units(plan, mods, xs)
nblk = K * 32 * 8 * 4
dbg = torch.zeros(nblk * 8, dtype=torch.int64, device="cuda")
for dma in (1, 0):
    lib.vam_resunit_set_dma(dma)
    for _ in range(3):
        plan.run()
    lib.vam_resunit_set_debug(dbg.data_ptr())
    plan.run(); torch.cuda.synchronize()
    lib.vam_resunit_set_debug(None)
    d = dbg.view(nblk, 8).cpu().double()
    ph = d[:, 1:7] - d[:, 0:6]
    names = ["GEMM1 (x + W1 staged)", "t1 write", "GEMM2 (27 items)", "t2 write", "GEMM3 (6 items)", "epilogue (issue)"]
    tot = (d[:, 6] - d[:, 0])
    print(f"dma={dma}: s_memtime ticks (100 MHz reference clock ticks if constant-rate) per workgroup, median over {nblk}: total {tot.median():.0f}")
    for n, col in zip(names, ph.t()):
        print(f"   {n:26s} median {col.median():9.0f}  p10 {col.quantile(0.1):9.0f}  p90 {col.quantile(0.9):9.0f}  ({100 * col.median() / tot.median():5.1f} %)")
    span = d[:, 6].max() - d[:, 0].min()
    print(f"   launch span {span:.0f} ticks; sum of workgroup times / 256 CUs = {tot.sum() / 256:.0f}")
lib.vam_resunit_set_dma(-1)
