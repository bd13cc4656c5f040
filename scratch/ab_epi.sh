#!/bin/bash
# direct vs staged epilogue on ONE box: ops tests first, then per-layer and whole-step timings, interleaved twice
cd /root/repo
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "epilogue or conv or deconv or gdn or attention or rem or subpel" > gpurun_out/epi_ops_test.log 2>&1 || { tail -30 gpurun_out/epi_ops_test.log; exit 1; }
tail -2 gpurun_out/epi_ops_test.log
for r in 1 2; do
  timeout -k 10 120 python scratch/diag_bench.py > gpurun_out/epi_direct_$r.log 2>&1
  VAMPIC_EPILOGUE=staged timeout -k 10 120 python scratch/diag_bench.py > gpurun_out/epi_staged_$r.log 2>&1
  timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/epi_bench_direct_$r.log 2>gpurun_out/epi_bench_direct_$r.err
  VAMPIC_EPILOGUE=staged timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/epi_bench_staged_$r.log 2>gpurun_out/epi_bench_staged_$r.err
done
python - <<'PY'
import re, json
rows = {}
for n in ("direct", "staged"):
    for r in (1, 2):
        for ln in open(f"gpurun_out/epi_{n}_{r}.log"):
            m = re.match(r"(.*?) tile (\S+)\s+([\d.]+) us", ln)
            if m: rows.setdefault((m.group(1), m.group(2)), {}).setdefault(n, []).append(float(m.group(3)))
print("%-44s %-8s %9s %9s %7s" % ("shape", "tile", "staged", "direct", "gain"))
for (s, t), d in rows.items():
    a, b = min(d["staged"]), min(d["direct"])
    print("%-44s %-8s %9.1f %9.1f %6.1f%%" % (s, t, a, b, 100 * (a / b - 1)))
for n in ("direct", "staged"):
    for r in (1, 2):
        try:
            j = json.loads(open(f"gpurun_out/epi_bench_{n}_{r}.log").read().strip().splitlines()[-1])
            print(n, r, j["ms_per_step"], "ms", j["roofline"]["classes"])
        except Exception as e:
            print(n, r, "failed", e)
PY
