import sys, torch, itertools, json
sys.path.insert(0, "/root/repo")
import vampic
from vampic import ops, layers as Ly, _lib as L
lib = L.load()
# (label, nprob, cin, n, k, stride, B, H, W)
SHAPES = [
 ("ga 5x5s2 192->192 @128", 2, 192, 192, 5, 2, 32, 128, 128),
 ("ga 5x5s2 192->192 @64", 2, 192, 192, 5, 2, 32, 64, 64),
 ("ga 5x5s2 192->320 @32", 2, 192, 320, 5, 2, 32, 32, 32),
 ("ru 3x3 96->96 @64 x4", 4, 96, 96, 3, 1, 32, 64, 64),
 ("ru 1x1 192->96 @64 x4", 4, 192, 96, 1, 1, 32, 64, 64),
 ("ru 1x1 96->192 @64 x4", 4, 96, 192, 1, 1, 32, 64, 64),
 ("gdn 1x1 192->192 @128 x2", 2, 192, 192, 1, 1, 32, 128, 128),
 ("qkv 1x1 192->576 @64 x2", 2, 192, 576, 1, 1, 32, 64, 64),
 ("deconv-phase 3x3 192->192 @64 x4", 4, 192, 192, 3, 1, 32, 64, 64),
 ("cc 3x3 512->224 @16 x2", 2, 512, 224, 3, 1, 32, 16, 16),
 ("cc 3x3 224->176 @16 x2", 2, 224, 176, 3, 1, 32, 16, 16),
 ("cc 3x3 176->128 @16 x2", 2, 176, 128, 3, 1, 32, 16, 16),
 ("cc 3x3 128->64 @16 x2", 2, 128, 64, 3, 1, 32, 16, 16),
 ("cc 3x3 64->32 @16 x2", 2, 64, 32, 3, 1, 32, 16, 16),
 ("cc 3x3 480->224 @16 x8", 8, 480, 224, 3, 1, 32, 16, 16),
 ("cc 3x3 224->176 @16 x8", 8, 224, 176, 3, 1, 32, 16, 16),
 ("cc 3x3 64->32 @16 x8", 8, 64, 32, 3, 1, 32, 16, 16),
 ("ru 3x3 160->160 @16 x4", 4, 160, 160, 3, 1, 32, 16, 16),
 ("ha 3x3 640->320 @16", 1, 640, 320, 3, 1, 32, 16, 16),
 ("first 3x3 16->192 @128 x2", 2, 16, 192, 3, 1, 32, 128, 128),
 ("last 3x3 192->12 @128", 1, 192, 12, 3, 1, 32, 128, 128),
]
res = {}
for (label, npb, cin, n, k, st, B, H, W) in SHAPES:
    probs = []
    keep = []
    for i in range(npb):
        m = Ly.Conv2d(cin, n, k, st).cuda()
        x = ops.new_view(B, H, W, cin); x.buf.normal_()
        Ho = H // st
        o = ops.new_view(B, Ho, Ho, n)
        probs.append(ops.conv_problem(m.packed(), [x], o, L.ACT_GELU)); keep += [m, x, o]
    flops = 2.0 * npb * B * (H // st) * (W // st) * n * cin * k * k
    best = []
    mode1 = lib.vam_conv_get_mode() == 1
    for bm, bn, bk in itertools.product((128, 64), (32, 64, 96, 128, 160, 192, 224), (32,) if mode1 else (16, 32)):
        if bk == 32 and cin % 32 and not mode1: continue
        if mode1 and (bm, bn) not in ((128,32),(128,64),(128,96),(128,128),(128,192),(64,32),(64,64),(64,128),(64,192)): continue
        npad = (n + 31) // 32 * 32
        if bn > npad and bn != 32: 
            if bn - npad >= 32: continue
        lib.vam_conv_force_tile(bm, bn, bk)
        try:
            ops.conv_group(probs); torch.cuda.synchronize()
        except Exception as e:
            continue
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(3):
            a.record()
            for _ in range(5): ops.conv_group(probs)
            b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 5)
        best.append((min(ts), bm, bn, bk))
    best.sort()
    lib.vam_conv_force_tile(0, 0, 0)
    ops.conv_group(probs); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): ops.conv_group(probs)
    b.record(); torch.cuda.synchronize(); auto = a.elapsed_time(b) / 5
    import ctypes
    cb = [ctypes.c_int() for _ in range(3)]
    lib.vam_conv_last_tile(*[ctypes.byref(c) for c in cb])
    print(f"{label:36s} auto[{cb[0].value}x{cb[1].value}x{cb[2].value}] {auto*1e3:8.1f}us {flops/auto/1e9:6.1f}TF | " + "  ".join(f"{bm}x{bn}x{bk}:{flops/t/1e9:.0f}" for t, bm, bn, bk in best[:6]), flush=True)
    del keep
