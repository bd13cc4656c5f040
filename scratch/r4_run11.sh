#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_wgrad_lds.py tests/test_gpu_first_train.py -q > gpurun_out/r4_t11.log 2>&1; tail -12 gpurun_out/r4_t11.log
for i in 1 2; do
  timeout -k 10 300 python scripts/bench_train.py --steps 5 --warmup 2 > gpurun_out/r4_p3on_$i.log 2>&1
  VAMPIC_TRAIN_P3=0 timeout -k 10 300 python scripts/bench_train.py --steps 5 --warmup 2 > gpurun_out/r4_p3off_$i.log 2>&1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4_p3o*_*.log')):
    l=[l for l in open(f) if l.startswith('{')]
    if l:
        d=json.loads(l[-1]); print(f, d['ms_per_step'], d['phase_ms'], d['config']['loss'])
    else: print(f, 'no result')
PY
