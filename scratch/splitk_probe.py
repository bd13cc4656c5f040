"""What split-K could buy on the slice-chain layers: time 2x[224->176] (today's launch) against 8x[64->176]
with the wide tile forced (= the conv part of a 4-way K split; the reduce pass is extra)."""
import sys, torch
sys.path.insert(0, "/root/repo")
import vampic
from vampic import ops, layers as Ly, _lib as L
lib = L.load()
def run(npb, cin, n, force=None):
    probs, keep = [], []
    for i in range(npb):
        m = Ly.Conv2d(cin, n, 3, 1).cuda()
        x = ops.new_view(32, 16, 16, cin); x.buf.normal_()
        o = ops.new_view(32, 16, 16, n)
        probs.append(ops.conv_problem(m.packed(), [x], o, L.ACT_GELU)); keep += [m, x, o]
    if force: lib.vam_conv_force_tile(*force)
    for _ in range(3): ops.conv_group(probs)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): ops.conv_group(probs)
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 10)
    lib.vam_conv_force_tile(0, 0, 0)
    return best * 1e3
for cin, n in ((224, 176), (176, 128), (512, 224)):
    t0 = run(2, cin, n)
    q = (cin // 4 + 31) // 32 * 32
    for tile in ((128, 192, 32), (128, 128, 32), (128, 96, 32), (64, 192, 32)):
        if tile[1] > (n + 31) // 32 * 32 + 31: continue
        t1 = run(8, q, n, tile)
        print(f"2x[{cin}->{n}] {t0:6.1f} us | 8x[{q}->{n}] tile {tile[0]}x{tile[1]}: {t1:6.1f} us (K padded to {4*q})")
