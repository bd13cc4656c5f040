"""Micro-benchmark of representative conv launches (TF/s with HIP events); A/B via VAMPIC_LIB=<other .so>."""
import sys, torch
sys.path.insert(0, "/root/repo")
import vampic
from vampic import ops, layers as Ly, _lib as L
SHAPES = [  # (n_problems, cin, n, k, stride, B, H, W)
    (2, 192, 192, 5, 2, 32, 128, 128), (4, 192, 192, 3, 1, 32, 64, 64), (4, 96, 96, 3, 1, 32, 64, 64),
    (2, 224, 176, 3, 1, 32, 16, 16), (2, 176, 128, 3, 1, 32, 16, 16), (2, 128, 64, 3, 1, 32, 16, 16), (2, 64, 32, 3, 1, 32, 16, 16),
    (2, 512, 224, 3, 1, 32, 16, 16), (8, 384, 224, 3, 1, 32, 16, 16), (8, 224, 176, 3, 1, 32, 16, 16),
    (4, 96, 192, 1, 1, 32, 64, 64), (4, 192, 96, 1, 1, 32, 64, 64), (2, 192, 192, 1, 1, 32, 128, 128), (2, 192, 576, 1, 1, 32, 64, 64),
]
import os
if os.environ.get("FORCE"):
    bm, bn, bk = (int(v) for v in os.environ["FORCE"].split(","))
    L.load().vam_conv_force_tile(bm, bn, bk)
    SHAPES = SHAPES[:3] + SHAPES[8:10]
for npb, cin, n, k, st, B, H, W in SHAPES:
    probs, keep = [], []
    for i in range(npb):
        m = Ly.Conv2d(cin, n, k, st).cuda()
        x = ops.new_view(B, H, W, cin); x.buf.normal_()
        o = ops.new_view(B, H // st, W // st, n)
        probs.append(ops.conv_problem(m.packed(), [x], o, L.ACT_GELU)); keep += [m, x, o]
    for _ in range(3):
        ops.conv_group(probs)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            ops.conv_group(probs)
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 10)
    fl = 2.0 * npb * B * (H // st) * (W // st) * n * cin * k * k
    print(f"{npb}x[{cin}->{n} k{k} s{st} P={B * (H // st) * (W // st)}]  {best * 1e3:8.1f} us  {fl / best / 1e9:6.1f} TF/s", flush=True)
