#!/bin/bash
# wave-specialised conv blocks (VAMPIC_SPEC=1) against the one-role kernel on ONE box: correctness first, then speed
cd /root/repo
VAMPIC_SPEC=1 timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -q -x > gpurun_out/spec_ops_test.log 2>&1
echo "ops tests with SPEC=1: rc=$?"; tail -3 gpurun_out/spec_ops_test.log
for r in 1 2; do
  VAMPIC_SPEC=0 timeout -k 10 120 python scratch/diag_bench.py > gpurun_out/spec_A_$r.log 2>&1
  VAMPIC_SPEC=1 timeout -k 10 120 python scratch/diag_bench.py > gpurun_out/spec_B_$r.log 2>&1
done
FORCE_TILE=64,192 VAMPIC_SPEC=0 timeout -k 10 120 python scratch/diag_bench.py > gpurun_out/spec_F0.log 2>&1
FORCE_TILE=64,192 VAMPIC_SPEC=1 timeout -k 10 120 python scratch/diag_bench.py > gpurun_out/spec_F1.log 2>&1
FORCE_TILE=128,128 VAMPIC_SPEC=1 timeout -k 10 120 python scratch/diag_bench.py > gpurun_out/spec_G1.log 2>&1
echo "--- forced 64x192 one-role"; cat gpurun_out/spec_F0.log | grep tile
echo "--- forced 64x192 spec"; cat gpurun_out/spec_F1.log | grep tile
echo "--- forced 128x128 spec"; cat gpurun_out/spec_G1.log | grep tile
python - <<'PY'
import re
rows={}
for n in "AB":
    for r in (1,2):
        for ln in open(f"gpurun_out/spec_{n}_{r}.log"):
            m=re.match(r"(.*?) tile (\S+)\s+([\d.]+) us",ln)
            if m: rows.setdefault((m.group(1),m.group(2)),{}).setdefault(n,[]).append(float(m.group(3)))
print("%-44s %-8s %9s %9s %7s"%("shape","tile","one-role","spec","ratio"))
for (s,t),d in rows.items():
    a,b=min(d.get("A",[0])),min(d.get("B",[0]))
    print("%-44s %-8s %9.1f %9.1f %7.2f"%(s,t,a,b,a/b if b else 0))
PY
for r in 1 2; do
  VAMPIC_SPEC=0 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/one-role /'
  VAMPIC_SPEC=1 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/spec     /'
done
