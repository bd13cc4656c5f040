import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
import vampic
from vampic import _lib as L, ops
L.LIB_PATH = "/root/repo/scratch/libvampic_dbg.so"
import vampic.synth as synth
s = synth.synth_sigma(3, 8192, seed=3).reshape(3, 32, 16, 16).cuda()
v = ops.from_nchw(s)
m = ops.new_view(v.B, v.H, v.W, v.C)
thr = torch.zeros(12, device="cuda")
ops.variance_mask(v, 0.01, m, n_slice=1, thr=thr)
np.set_printoptions(precision=9)
print(thr.cpu().numpy().reshape(3,4))
srt = torch.sort(s.reshape(3,-1), dim=1).values.cpu().numpy()
print(srt[:, 8181:8185])
