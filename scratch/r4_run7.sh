#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python scripts/bench_train.py --steps 5 --warmup 2 > gpurun_out/r4_train5.log 2>&1; tail -1 gpurun_out/r4_train5.log | cut -c1-500
timeout -k 10 700 python -m pytest tests/test_gpu_first_train.py tests/test_gpu_train_gs.py tests/test_gpu_train.py -q > gpurun_out/r4_t7.log 2>&1; tail -8 gpurun_out/r4_t7.log
for i in 1 2; do
  timeout -k 10 200 python bench.py --no-train --no-bf16 --no-cpu-baseline --steps 20 > gpurun_out/r4_nt0_$i.log 2>&1
  VAMPIC_NT_STORE=1 timeout -k 10 200 python bench.py --no-train --no-bf16 --no-cpu-baseline --steps 20 > gpurun_out/r4_nt1_$i.log 2>&1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4_nt*_*.log')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); r=d['roofline']
    print(f, d['ms_per_step'], r['frac'], r['g_a_g_s']['frac'], r['g_a_g_s']['ms_per_step'])
PY
