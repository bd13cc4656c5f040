#!/bin/bash
cd /root/repo
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -q -x > gpurun_out/spec8_ops_test.log 2>&1
echo "ops tests (auto SPEC, 8-consumer large tiles): rc=$?"; tail -2 gpurun_out/spec8_ops_test.log
for r in 1 2; do
  VAMPIC_SPEC=0 timeout -k 10 120 python scratch/diag_bench.py > gpurun_out/s8_A_$r.log 2>&1
  VAMPIC_SPEC8=0 timeout -k 10 120 python scratch/diag_bench.py > gpurun_out/s8_B_$r.log 2>&1
  timeout -k 10 120 python scratch/diag_bench.py > gpurun_out/s8_C_$r.log 2>&1
done
python - <<'PY'
import re
rows={}
for n in "ABC":
    for r in (1,2):
        for ln in open(f"gpurun_out/s8_{n}_{r}.log"):
            m=re.match(r"(.*?) tile (\S+)\s+([\d.]+) us",ln)
            if m: rows.setdefault((m.group(1),m.group(2)),{}).setdefault(n,[]).append(float(m.group(3)))
print("%-44s %-8s %9s %9s %9s"%("shape","tile","one-role","spec4+4","spec8+4"))
for (s,t),d in rows.items():
    print("%-44s %-8s %9.1f %9.1f %9.1f"%(s,t,min(d.get("A",[0])),min(d.get("B",[0])),min(d.get("C",[0]))))
PY
for r in 1 2; do
  VAMPIC_SPEC=0 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed 's/^/one-role /'
  VAMPIC_SPEC8=0 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed 's/^/spec 4+4 /'
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed 's/^/spec 8+4 /'
done
