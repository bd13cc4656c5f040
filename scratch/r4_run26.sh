#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_first_train.py -q -x > gpurun_out/r4_t26.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r4_t26.log
for i in 1 2; do
  for m in 0 1; do
    VAMPIC_AXPY_GROUP=$m timeout -k 10 300 python scripts/bench_train.py --steps 5 --warmup 2 > gpurun_out/r4_ax${m}_$i.log 2>&1
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4_ax*_*.log')):
    l=[l for l in open(f) if l.startswith('{')]
    if l:
        d=json.loads(l[-1]); print(f, d['ms_per_step'], d['phase_ms'], d['config']['loss'])
    else: print(f, 'no result')
PY
