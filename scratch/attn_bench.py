import sys, torch
sys.path.insert(0, "/root/repo")
import vampic
from vampic import ops
for (B, H, W, C, ws, sh) in ((32, 64, 64, 192, 8, 4), (32, 16, 16, 320, 4, 2)):
    qkv = ops.new_view(B, H, W, 3 * C); qkv.buf.normal_()
    out = ops.new_view(B, H, W, C)
    tab = torch.randn(((2 * ws - 1) ** 2, 8), device="cuda")
    for _ in range(3): ops.win_attention(qkv, out, tab, C, 8, ws, sh)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): ops.win_attention(qkv, out, tab, C, 8, ws, sh)
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 10)
    print(f"attn B{B} {H}x{W} C{C} ws{ws}: {best*1e3:.1f} us  checksum {out.buf.double().sum().item():.6f}")
