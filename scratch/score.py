import re, itertools, math
SHAPES = {}
exec(open("/root/repo/scratch/tune.py").read().split("res = {}")[0].split("lib = L.load()")[1])
def cdiv(a,b): return (a+b-1)//b
SHAPE_PEN = {(128,192):1.0,(128,224):1.06,(128,160):1.04,(128,128):1.04,(128,96):1.07,(128,64):1.07,(128,32):1.35,
             (64,192):1.06,(64,128):1.07,(64,64):1.13,(64,96):1.3,(64,160):1.35,(64,224):1.6,(64,32):1.7}
def choose(npb, cin, n, k, P):
    npad = cdiv(n,32)*32; ktot = cin*k*k
    best=None
    for (bm,bn),sp in SHAPE_PEN.items():
        tiles_n = cdiv(npad,bn); blocks = npb*cdiv(P,bm)*tiles_n
        waste = tiles_n*bn/n * (cdiv(P,bm)*bm/P)
        if blocks >= 1024: bp=1.0
        elif blocks >= 512: bp=1.04
        elif blocks >= 256: bp=1.22
        else: bp=1.22*256/blocks
        kp = 1.0
        if ktot <= 256 and bn > 96: kp = 1.3
        if ktot <= 256 and bn == 96: kp = 1.1
        sc = waste*bp*sp*kp
        if best is None or sc < best[0]: best=(sc,bm,bn)
    _,bm,bn = best
    bk = 32 if cin%32==0 else 16
    if (bm,bn)==(128,192) or ktot<=256: bk=16
    return bm,bn,bk
log = open("/root/repo/gpurun_out/tune1.log").read().splitlines()
for (label, npb, cin, n, k, st, B, H, W) in SHAPES:
    P = B*(H//st)*(W//st)
    c = choose(npb,cin,n,k,P)
    line = [l for l in log if l.startswith(label)][0]
    print(f"{label:36s} -> {c[0]}x{c[1]}x{c[2]:<3d} | {line.split('|')[1][:70]}")
