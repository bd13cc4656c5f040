import sys, torch
sys.path.insert(0, "/root/repo")
import vampic
from vampic import ops, layers as Ly, _lib as L
lib = L.load()
def run(label, npb, cin, n, k, st, B, H, W, force=(0,0,0)):
    probs, keep = [], []
    for i in range(npb):
        m = Ly.Conv2d(cin, n, k, st).cuda()
        x = ops.new_view(B, H, W, cin); x.buf.normal_()
        o = ops.new_view(B, H // st, W // st, n)
        probs.append(ops.conv_problem(m.packed(), [x], o, L.ACT_NONE)); keep += [m, x, o]
    flops = 2.0 * npb * B * (H // st) * (W // st) * n * cin * k * k
    lib.vam_conv_force_tile(*force)
    ops.conv_group(probs); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        a.record()
        for _ in range(3): ops.conv_group(probs)
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 3)
    print(f"{label:44s} {min(ts)*1e3:9.1f} us {flops/min(ts)/1e9:7.1f} TF", flush=True)
    lib.vam_conv_force_tile(0, 0, 0)
run("conv5x5s2 192->192 @128 x2 (ref)", 2, 192, 192, 5, 2, 32, 128, 128)
run("GEMM 1x1 K=4800 N=192 P=65536 x4", 4, 4800, 192, 1, 1, 16, 64, 64, (128, 192, 16))
run("GEMM 1x1 K=4800 N=192 P=65536 x4 bk32", 4, 4800, 192, 1, 1, 16, 64, 64, (128, 192, 32))
run("GEMM 1x1 K=1728 N=192 P=131072 x4", 4, 1728, 192, 1, 1, 32, 64, 64, (128, 192, 16))
run("GEMM 1x1 K=1728 N=128 P=131072 x4 128x128x32", 4, 1728, 128, 1, 1, 32, 64, 64, (128, 128, 32))
run("conv3x3 192->192 @64 x4 (ref)", 4, 192, 192, 3, 1, 32, 64, 64)
