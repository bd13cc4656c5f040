"""The LDS-tiled weight-gradient kernel (csrc/wgrad_lds.hip, round 4): k3 / k5 layers on grids of width 16 / 32 / 64k,
stride 1 and 2, against autograd of F.conv2d (fp32 CPU; tolerance 2e-5 of the max: different summation order), against
the register-gather kernel it replaces (VAMPIC_WGRAD_LDS=0 in a child process is not needed: shapes the new kernel
declines — width 8, 1x1 — still take the old one inside the same launch group), and bit-reproducible."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import vampic.synth as synth               # noqa: E402
from vampic import _lib as L, ops          # noqa: E402


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


CASES = [
    # k, stride, segments, N, (H, W) of the OUTPUT grid, B
    (3, 1, (64,), 128, (16, 16), 2),            # latent grid, one full tile
    (3, 1, (320, 32, 64), 224, (16, 16), 4),    # stack head: three segments, N = 224 (ragged second n tile)
    (3, 1, (224,), 176, (16, 16), 2),           # N = 176, C = 224 (ragged tiles both ways)
    (3, 1, (96,), 96, (32, 32), 2),             # width 32: two image rows per chunk
    (3, 1, (96,), 96, (64, 64), 1),             # width 64: one row per chunk
    (3, 1, (16,), 192, (64, 128), 1),           # first layer (space-to-depth input, 16 channels), width 128
    (5, 2, (192,), 192, (16, 16), 2),           # k5 stride 2: x is 32 x 32
    (5, 2, (192,), 320, (32, 32), 1),
    (5, 2, (64,), 64, (64, 64), 1),             # k5 stride 2, width 64: x is 128 x 128
    (3, 2, (64,), 64, (16, 16), 2),             # k3 stride 2
    (5, 1, (32,), 64, (16, 32), 2),             # k5 stride 1
    (3, 1, (64,), 64, (16, 16), 40),            # 10240 pixels: pixel splits + reduce
    (1, 1, (192,), 192, (64, 64), 1),           # 1x1 layers: waves own 2 x 1 / 1 x 2 / 2 x 2 blocks, 32-pixel chunks
    (1, 1, (96,), 192, (16, 16), 4),
    (1, 1, (192,), 96, (32, 32), 2),
    (1, 1, (160, 32), 320, (16, 16), 2),
    (1, 1, (192,), 576, (16, 16), 12),          # 3072 pixels: splits
]


@pytest.mark.parametrize("k,stride,segs,n,hw,B", CASES)
def test_lds_wgrad_matches_autograd(k, stride, segs, n, hw, B):
    H, W = hw
    cin = sum(segs)
    Hx, Wx = H * stride, W * stride
    xs = [synth.normal((B, c, Hx, Wx), 3 + i) for i, c in enumerate(segs)]
    dy = synth.normal((B, n, H, W), 9)
    w = synth.normal((n, cin, k, k), 1, 0.1).requires_grad_(True)
    b = synth.normal((n,), 2, 0.1).requires_grad_(True)
    F.conv2d(torch.cat(xs, 1), w, b, stride=stride, padding=k // 2).backward(dy)
    xv = [ops.from_nchw(t.cuda()) for t in xs]
    dyv = ops.from_nchw(dy.cuda())
    dw = torch.full((n, cin, k, k), float("nan"), device="cuda")
    db = torch.full((n,), float("nan"), device="cuda")
    probs = ops.wgrad_problems(xv, dyv, dw, db, stride=stride)
    if B * H * W >= 8192:
        assert probs[0].splits > 1
    ops.wgrad_group(probs)
    torch.cuda.synchronize()
    assert _rel(dw, w.grad) <= 2e-5, _rel(dw, w.grad)
    assert _rel(db, b.grad) <= 2e-5
    dw2, db2 = torch.full_like(dw, float("nan")), torch.full_like(db, float("nan"))
    ops.wgrad_group(ops.wgrad_problems(xv, dyv, dw2, db2, stride=stride))
    torch.cuda.synchronize()
    assert torch.equal(dw, dw2) and torch.equal(db, db2)          # fixed accumulation order


def test_mixed_group_routes_each_problem_to_its_kernel():
    """One grouped launch holding a k3 problem on a 16 x 16 grid (LDS-tiled kernel), one on an 8 x 8 grid and a 1x1 layer
    (register-gather kernel): every problem gets its own result."""
    out = []
    probs = []
    for (k, hw, c, n) in [(3, (16, 16), 64, 64), (3, (8, 8), 64, 64), (1, (16, 16), 96, 192), (1, (8, 8), 96, 192)]:
        x = synth.normal((2, c, *hw), 11 + k + hw[0])
        dy = synth.normal((2, n, *hw), 12 + k + hw[0])
        w = synth.normal((n, c, k, k), 1, 0.1).requires_grad_(True)
        F.conv2d(x, w, None, padding=k // 2).backward(dy)
        dw = torch.full((n, c, k, k), float("nan"), device="cuda")
        db = torch.full((n,), float("nan"), device="cuda")
        xv, dyv = ops.from_nchw(x.cuda()), ops.from_nchw(dy.cuda())      # (the problem structs hold raw pointers: keep the views)
        probs += ops.wgrad_problems([xv], dyv, dw, db)
        out.append((dw, w.grad, db, dy.sum((0, 2, 3)), xv, dyv))
    ops.wgrad_group(probs)
    torch.cuda.synchronize()
    for dw, ref, db, refb, _, _ in out:
        assert _rel(dw, ref) <= 2e-5 and _rel(db, refb) <= 2e-5


@pytest.mark.parametrize("segs,n,hw,B", [((224,), 176, (16, 16), 2), ((64,), 64, (32, 32), 1), ((32,), 64, (16, 16), 2), ((96,), 96, (16, 16), 2)])
def test_plane_input_gives_the_same_bits_as_fp32_input(segs, n, hw, B):
    """X handed over as bf16x3 planes (the taped activations of the slice stacks, written by the producing launch's
    epilogue): the staged tile holds the same three bf16 terms as the in-kernel split of the fp32 tensor, so the weight
    gradient is bit-identical."""
    from vampic import engine as E
    H, W = hw
    c = segs[0]
    x = synth.normal((B, c, H, W), 3)
    dy = synth.normal((B, n, H, W), 9)
    xv, dyv = ops.from_nchw(x.cuda()), ops.from_nchw(dy.cuda())
    x3 = ops.new_view3(B, H, W, c)
    ops.conv_group([ops.conv_problem(E._identity_pack(c, "cuda"), [xv], x3)])       # exact: 1.0 * x, then the epilogue's split
    assert torch.equal(x3.to_float().cpu(), xv.buf.cpu())
    assert ops.wgrad_reads_planes(H, W)
    res = []
    for inp in (xv, x3):
        dw = torch.full((n, c, 3, 3), float("nan"), device="cuda")
        db = torch.full((n,), float("nan"), device="cuda")
        probs = ops.wgrad_problems([inp], dyv, dw, db)
        assert bool(probs[0].flags & L.WGRAD_X_P3) == (inp is x3)
        ops.wgrad_group(probs)
        torch.cuda.synchronize()
        res.append((dw, db))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    w = synth.normal((n, c, 3, 3), 1, 0.1).requires_grad_(True)
    F.conv2d(x, w, None, padding=1).backward(dy)
    assert _rel(res[1][0], w.grad) <= 2e-5
