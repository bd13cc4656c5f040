import sys, torch
sys.path.insert(0, "/root/repo")
import vampic
from vampic import ops, layers as Ly, _lib as L
SHAPES = [(2, 224, 176, 3, 1, 32, 16, 16), (2, 176, 128, 3, 1, 32, 16, 16), (2, 128, 64, 3, 1, 32, 16, 16), (2, 64, 32, 3, 1, 32, 16, 16),
          (8, 224, 176, 3, 1, 32, 16, 16), (4, 96, 96, 3, 1, 32, 64, 64), (4, 96, 192, 1, 1, 32, 64, 64), (4, 192, 192, 3, 1, 32, 64, 64)]
def timeit(probs):
    for _ in range(3): ops.conv_group(probs)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): ops.conv_group(probs)
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 10)
    return best
for npb, cin, n, k, st, B, H, W in SHAPES:
    res = []
    for fmt_in, fmt_out in ((0, 0), (1, 0), (0, 1), (1, 1)):
        probs, keep = [], []
        for i in range(npb):
            m = Ly.Conv2d(cin, n, k, st).cuda()
            x = ops.new_view3(B, H, W, cin) if fmt_in else ops.new_view(B, H, W, cin)
            x.buf.normal_()
            if fmt_in: x.buf.copy_(torch.randint(0, 2**15, x.buf.shape, device="cuda").to(torch.float32).view(torch.int32).bitwise_and(0x3F803F80).view(torch.float32))
            o = ops.new_view3(B, H // st, W // st, n) if fmt_out else ops.new_view(B, H // st, W // st, n)
            probs.append(ops.conv_problem(m.packed(), [x], o, L.ACT_GELU)); keep += [m, x, o]
        res.append(timeit(probs))
    fl = 2.0 * npb * B * (H // st) * (W // st) * n * cin * k * k
    print(f"{npb}x[{cin}->{n} k{k} P={B*H*W}]  f32->f32 {res[0]*1e3:7.1f}us {fl/res[0]/1e9:6.1f}TF | P3 in {res[1]*1e3:7.1f}us {fl/res[1]/1e9:6.1f}TF | P3 out {res[2]*1e3:7.1f}us | P3 in+out {res[3]*1e3:7.1f}us {fl/res[3]/1e9:6.1f}TF", flush=True)
