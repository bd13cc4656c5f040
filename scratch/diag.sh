#!/bin/bash
# one box: the product library and every ablation build, interleaved twice.  The ablation builds came from temporary
# `#if defined(VAM_DIAG) && (VAM_DIAG & bit)` switches in csrc/conv_igemm.hip (bit 1: skip the B ds_write, 2: the A ds_write,
# 4: the MFMAs, 8: the B global loads, 16: the A global loads); they were removed again after the measurement
# (git history: commit "bf16-storage configuration" still carries them).
cd /root/repo
for r in 1 2; do
  timeout -k 10 120 python scratch/diag_bench.py > gpurun_out/diag_base_$r.log 2>&1
  for d in 1 2 3 4 8 16 24 27; do
    VAMPIC_LIB=/root/repo/scratch/libvampic_d$d.so timeout -k 10 120 python scratch/diag_bench.py > gpurun_out/diag_d${d}_$r.log 2>&1
  done
done
python - <<'PY'
import re,glob
names=["base"]+[f"d{d}" for d in (1,2,3,4,8,16,24,27)]
rows={}
for n in names:
    for r in (1,2):
        for ln in open(f"gpurun_out/diag_{n}_{r}.log"):
            m=re.match(r"(.*?) tile (\S+)\s+([\d.]+) us",ln)
            if m: rows.setdefault((m.group(1),m.group(2)),{}).setdefault(n,[]).append(float(m.group(3)))
print("%-44s %-8s"%("shape","tile")+"".join("%8s"%n for n in names))
for (s,t),d in rows.items():
    print("%-44s %-8s"%(s,t)+"".join("%8.1f"%min(d.get(n,[0])) for n in names))
PY
