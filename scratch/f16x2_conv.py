"""fp16x2 prototype (VAMPIC_CONV=f16x2): error of single conv launches against a float64 reference, and TF/s of the
representative launches with the input-max cells filled once outside the timed loop.  Run once per mode:
    python scratch/f16x2_conv.py            # bf16x3 (default)
    VAMPIC_CONV=f16x2 python scratch/f16x2_conv.py
"""
import os, sys, torch
sys.path.insert(0, "/root/repo")
import vampic
from vampic import ops, layers as Ly, _lib as L
import torch.nn.functional as F

lib = L.load()
mode = lib.vam_conv_get_mode()
print("conv mode", mode, flush=True)
torch.manual_seed(0)


def launch(probs):
    chunk = list(probs)
    keep = None
    if ops.f16x2_mode():
        keep = ops.amax_prepare(chunk)
        keep[1]()
    arr = (L.VamConv * len(chunk))(*chunk)
    return arr, len(chunk), keep


# ---- accuracy on small problems (float64 reference on the CPU)
for cin, n, k, st, B, H, W, act, gain in [(192, 192, 3, 1, 2, 32, 32, L.ACT_NONE, 1.0), (192, 192, 5, 2, 2, 64, 64, L.ACT_NONE, 1.0),
                                          (192, 192, 1, 1, 2, 32, 32, L.ACT_GELU, 1.0), (224, 176, 3, 1, 4, 16, 16, L.ACT_NONE, 1e-3),
                                          (64, 32, 3, 1, 4, 16, 16, L.ACT_NONE, 300.0), (320, 224, 3, 1, 2, 16, 16, L.ACT_NONE, 1.0)]:
    m = Ly.Conv2d(cin, n, k, st).cuda()
    with torch.no_grad():
        m.bias.normal_(0, 0.1)
        m.weight.mul_(torch.exp(torch.randn(n, 1, 1, 1, device="cuda")))     # per-channel weight scales over a few octaves
    x = ops.new_view(B, H, W, cin)
    # activations spanning several decades inside one tensor
    x.buf.copy_(torch.randn_like(x.buf) * torch.exp(2.0 * torch.randn(B, H, W, 1, device="cuda")) * gain)
    o = ops.new_view(B, H // st, W // st, n)
    arr, cnt, keep = launch([ops.conv_problem(m.packed(), [x], o, act)])
    L.check(lib.vam_conv_group(arr, cnt, ops.stream_ptr()), "conv")
    torch.cuda.synchronize()
    ref = F.conv2d(x.buf.permute(0, 3, 1, 2).double().cpu(), m.weight.double().cpu(), m.bias.double().cpu(), stride=st, padding=k // 2)
    if act == L.ACT_GELU:
        ref = F.gelu(ref)
    got = o.buf.permute(0, 3, 1, 2).double().cpu()
    ref32 = F.conv2d(x.buf.permute(0, 3, 1, 2).cpu(), m.weight.cpu(), m.bias.cpu(), stride=st, padding=k // 2).double()
    if act == L.ACT_GELU:
        ref32 = F.gelu(ref32)
    rms = ref.pow(2).mean().sqrt()
    e, e32 = (got - ref).abs(), (ref32 - ref).abs()
    print(f"[{cin}->{n} k{k} s{st} gain {gain:g}]  kernel: max|err|/rms {float(e.max() / rms):.3e} rms err/rms {float(e.pow(2).mean().sqrt() / rms):.3e}"
          f"   | CPU fp32 conv: max {float(e32.max() / rms):.3e} rms {float(e32.pow(2).mean().sqrt() / rms):.3e}", flush=True)

# ---- speed
SHAPES = [  # (n_problems, cin, n, k, stride, B, H, W)
    (2, 192, 192, 5, 2, 32, 128, 128), (4, 192, 192, 3, 1, 32, 64, 64), (4, 96, 96, 3, 1, 32, 64, 64),
    (2, 224, 176, 3, 1, 32, 16, 16), (2, 176, 128, 3, 1, 32, 16, 16), (2, 128, 64, 3, 1, 32, 16, 16), (2, 64, 32, 3, 1, 32, 16, 16),
    (8, 320, 224, 3, 1, 32, 16, 16), (8, 224, 176, 3, 1, 32, 16, 16),
    (4, 96, 192, 1, 1, 32, 64, 64), (4, 192, 96, 1, 1, 32, 64, 64), (2, 192, 192, 1, 1, 32, 128, 128), (2, 192, 576, 1, 1, 32, 64, 64),
    (2, 16, 192, 3, 1, 32, 128, 128),
]
for npb, cin, n, k, st, B, H, W in SHAPES:
    probs, keep = [], []
    for i in range(npb):
        m = Ly.Conv2d(cin, n, k, st).cuda()
        x = ops.new_view(B, H, W, cin); x.buf.normal_()
        o = ops.new_view(B, H // st, W // st, n)
        probs.append(ops.conv_problem(m.packed(), [x], o, L.ACT_GELU)); keep += [m, x, o]
    arr, cnt, kp = launch(probs)
    s = ops.stream_ptr()
    for _ in range(3):
        L.check(lib.vam_conv_group(arr, cnt, s), "conv")
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            lib.vam_conv_group(arr, cnt, s)
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 10)
    fl = 2.0 * npb * B * (H // st) * (W // st) * n * cin * k * k
    bm, bn, bk = (__import__("ctypes").c_int() for _ in range(3))
    lib.vam_conv_last_tile(__import__("ctypes").byref(bm), __import__("ctypes").byref(bn), __import__("ctypes").byref(bk))
    tabs = ""
    if kp is not None:       # cost of the prototype's input-max pass
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            kp[1]()
        b.record(); torch.cuda.synchronize()
        tabs = f"  (+ absmax pass {a.elapsed_time(b) * 100:.1f} us)"
    print(f"{npb}x[{cin}->{n} k{k} s{st} P={B * (H // st) * (W // st)}] tile {bm.value}x{bn.value}  {best * 1e3:8.1f} us  {fl / best / 1e9:6.1f} TF/s{tabs}", flush=True)
