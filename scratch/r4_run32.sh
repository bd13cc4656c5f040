#!/bin/bash
mkdir -p gpurun_out
VAMPIC_WGRAD_BRANCHES=2 timeout -k 10 400 python -m pytest tests/test_gpu_first_train.py -q -x -k "matches_reference and not variants or single_decoder" > gpurun_out/r4_t32.log 2>&1; echo "tests(2 branches) rc=$?"; tail -3 gpurun_out/r4_t32.log
for i in 1 2; do
  for m in 1 2 3; do
    VAMPIC_WGRAD_BRANCHES=$m timeout -k 10 300 python scripts/bench_train.py --steps 5 --warmup 2 > gpurun_out/r4_wb${m}_$i.log 2>&1
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4_wb*_*.log')):
    l=[l for l in open(f) if l.startswith('{')]
    if l:
        d=json.loads(l[-1]); print(f, d['ms_per_step'], d['phase_ms'], d['config']['loss'])
    else: print(f, 'no result')
PY
