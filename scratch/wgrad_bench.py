"""Micro-benchmark of the weight-gradient launches of a first_train step (32x256x256): TF/s per distinct problem shape with
HIP events.  A/B: VAMPIC_WGRAD_LDS=0 (register-gather kernel) vs default (LDS-tiled kernel, csrc/wgrad_lds.hip)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vampic
from vampic import ops, _lib as L
SHAPES = [  # (n_problems, cin, n, k, stride, B, H, W of the OUTPUT grid, launches per step)
    (1, 192, 192, 5, 2, 32, 64, 64, 4), (1, 192, 192, 5, 2, 32, 32, 32, 4), (1, 192, 320, 5, 2, 32, 16, 16, 4),
    (1, 16, 192, 3, 1, 32, 128, 128, 2),
    (1, 96, 96, 3, 1, 32, 64, 64, 24), (1, 160, 160, 3, 1, 32, 16, 16, 24),
    (6, 320, 224, 3, 1, 32, 16, 16, 9), (16, 320, 224, 3, 1, 32, 16, 16, 2), (2, 224, 176, 3, 1, 32, 16, 16, 15),
    (2, 176, 128, 3, 1, 32, 16, 16, 15), (2, 128, 64, 3, 1, 32, 16, 16, 15), (2, 64, 32, 3, 1, 32, 16, 16, 15),
    (16, 32, 224, 3, 1, 32, 16, 16, 1), (10, 224, 176, 3, 1, 32, 16, 16, 2),
    (1, 192, 192, 1, 1, 32, 64, 64, 12), (1, 192, 192, 1, 1, 32, 128, 128, 8), (1, 96, 192, 1, 1, 32, 64, 64, 24),
    (1, 192, 96, 1, 1, 32, 64, 64, 24), (1, 192, 576, 1, 1, 32, 64, 64, 4), (1, 160, 320, 1, 1, 32, 16, 16, 24), (1, 320, 160, 1, 1, 32, 16, 16, 24),
]
only = os.environ.get("ONLY")
tot = 0.0
for npb, cin, n, k, st, B, H, W, per_step in SHAPES:
    if only and str(k) not in only.split(","):
        continue
    probs, keep = [], []
    for i in range(npb):
        x = ops.new_view(B, H * st, W * st, cin); x.buf.normal_()
        dy = ops.new_view(B, H, W, n); dy.buf.normal_()
        dw = torch.empty((n, cin, k, k), device="cuda"); db = torch.empty((n,), device="cuda")
        probs += ops.wgrad_problems([x], dy, dw, db, stride=st); keep += [x, dy, dw, db]
    for _ in range(2):
        ops.wgrad_group(probs)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            ops.wgrad_group(probs)
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 5)
    fl = 2.0 * npb * B * H * W * n * cin * k * k
    tot += best * per_step
    print(f"{npb:2d}x[{cin}->{n} k{k} s{st} P={B * H * W}] splits {probs[0].splits:3d} {best * 1e3:8.1f} us {fl / best / 1e9:6.1f} TF/s  x{per_step} = {best * per_step:6.2f} ms/step", flush=True)
print(f"sum over the listed launches: {tot:.2f} ms per step")
