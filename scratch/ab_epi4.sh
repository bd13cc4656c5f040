#!/bin/bash
cd /root/repo
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x > gpurun_out/epi4_ops_test.log 2>&1 || { tail -30 gpurun_out/epi4_ops_test.log; exit 1; }
tail -2 gpurun_out/epi4_ops_test.log
for r in 1 2 3; do
  timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/epi4_bench_direct_$r.log 2>gpurun_out/epi4_bench_direct_$r.err
  VAMPIC_EPILOGUE=staged timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/epi4_bench_staged_$r.log 2>gpurun_out/epi4_bench_staged_$r.err
done
python - <<'PY'
import json
for n in ("direct", "staged"):
    for r in (1, 2, 3):
        j = json.loads(open(f"gpurun_out/epi4_bench_{n}_{r}.log").read().strip().splitlines()[-1])
        print(n, r, j["ms_per_step"], "ms", {k: v["ms_per_step"] for k, v in j["roofline"]["classes"].items()})
PY
