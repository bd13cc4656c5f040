import sys, torch
sys.path.insert(0, "/root/repo")
import vampic
from vampic import ops, layers as Ly, _lib as L
npb, cin, n, k, st, B, H, W, reps = [int(a) for a in sys.argv[1:10]]
probs, keep = [], []
for i in range(npb):
    m = Ly.Conv2d(cin, n, k, st).cuda()
    x = ops.new_view(B, H, W, cin); x.buf.normal_()
    o = ops.new_view(B, H // st, W // st, n)
    probs.append(ops.conv_problem(m.packed(), [x], o, L.ACT_GELU)); keep += [m, x, o]
for _ in range(reps):
    ops.conv_group(probs)
torch.cuda.synchronize()
