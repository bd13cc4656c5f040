#!/bin/bash
cd /root/repo
for r in 1 2; do
 for act in ACT_NONE ACT_LEAKY ACT_GELU; do
  ACT=$act timeout -k 10 120 python scratch/diag_bench.py > gpurun_out/epi2_direct_${act}_$r.log 2>&1
  ACT=$act VAMPIC_EPILOGUE=staged timeout -k 10 120 python scratch/diag_bench.py > gpurun_out/epi2_staged_${act}_$r.log 2>&1
 done
done
python - <<'PY'
import re
rows = {}
for act in ("ACT_NONE","ACT_LEAKY","ACT_GELU"):
  for n in ("direct", "staged"):
    for r in (1, 2):
        for ln in open(f"gpurun_out/epi2_{n}_{act}_{r}.log"):
            m = re.match(r"(.*?) tile (\S+)\s+([\d.]+) us", ln)
            if m: rows.setdefault((m.group(1), m.group(2)), {}).setdefault((act,n), []).append(float(m.group(3)))
print("%-40s %-8s" % ("shape", "tile") + "".join("%9s" % (a[4:8]+"/"+n[:3]) for a in ("ACT_NONE","ACT_LEAKY","ACT_GELU") for n in ("staged","direct")))
for (s, t), d in rows.items():
    print("%-40s %-8s" % (s, t) + "".join("%9.1f" % min(d[(a,n)]) for a in ("ACT_NONE","ACT_LEAKY","ACT_GELU") for n in ("staged","direct")))
PY
