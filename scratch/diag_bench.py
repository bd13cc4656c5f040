"""Ablation bench: time representative conv launches (fp32 and bf16x3-plane inputs) for the library named by VAMPIC_LIB.
Timing-only builds (scratch/libvampic_d*.so, -DVAM_DIAG=bits: 1 no B ds_write, 2 no A ds_write, 4 no MFMA, 8 no B global
load, 16 no A global load) compute garbage; only the time matters."""
import os, sys, torch
sys.path.insert(0, "/root/repo")
import vampic
from vampic import ops, layers as Ly, _lib as L
import ctypes as C
SHAPES = [  # (n_problems, cin, n, k, stride, B, H, W, p3_in)
    (2, 192, 192, 5, 2, 32, 128, 128, 0), (4, 192, 192, 3, 1, 32, 64, 64, 0), (4, 96, 96, 3, 1, 32, 64, 64, 0),
    (8, 320, 224, 3, 1, 32, 16, 16, 0),
    (2, 224, 176, 3, 1, 32, 16, 16, 1), (2, 176, 128, 3, 1, 32, 16, 16, 1), (2, 128, 64, 3, 1, 32, 16, 16, 1), (2, 64, 32, 3, 1, 32, 16, 16, 1),
    (2, 192, 224, 3, 1, 32, 16, 16, 0),
    (4, 96, 192, 1, 1, 32, 64, 64, 0), (4, 192, 96, 1, 1, 32, 64, 64, 0), (2, 192, 192, 1, 1, 32, 128, 128, 0), (2, 192, 576, 1, 1, 32, 64, 64, 0),
    (2, 16, 192, 3, 1, 32, 128, 128, 0),
]
lib = L.load()
if os.environ.get("FORCE_TILE"):
    bm_, bn_ = (int(v) for v in os.environ["FORCE_TILE"].split(","))
    lib.vam_conv_force_tile(bm_, bn_, 32)
    SHAPES = [sh for sh in SHAPES if sh[2] % bn_ == 0 or sh[2] > bn_][:4] + SHAPES[11:12]
for npb, cin, n, k, st, B, H, W, p3 in SHAPES:
    probs, keep = [], []
    for i in range(npb):
        m = Ly.Conv2d(cin, n, k, st).cuda()
        if p3:
            x = ops.new_view3(B, H, W, cin); x.buf.view(torch.int16).random_(0x3c00, 0x4000)
        else:
            x = ops.new_view(B, H, W, cin); x.buf.normal_()
        o = ops.new_view(B, H // st, W // st, n)
        probs.append(ops.conv_problem(m.packed(), [x], o, getattr(L, os.environ.get('ACT', 'ACT_GELU')))); keep += [m, x, o]
    for _ in range(3):
        ops.conv_group(probs)
    torch.cuda.synchronize()
    bm, bn, bk = C.c_int(), C.c_int(), C.c_int()
    lib.vam_conv_last_tile(C.byref(bm), C.byref(bn), C.byref(bk))
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            ops.conv_group(probs)
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 10)
    fl = 2.0 * npb * B * (H // st) * (W // st) * n * cin * k * k
    print(f"{npb}x[{cin}->{n} k{k} s{st} P={B * (H // st) * (W // st)}{' P3' if p3 else ''}] tile {bm.value}x{bn.value} {best * 1e3:8.1f} us  {fl / best / 1e9:6.1f} TF/s", flush=True)
