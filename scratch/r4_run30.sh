#!/bin/bash
mkdir -p gpurun_out
for i in 1 2; do
  for m in 1.0 1.6 2.5; do
    VAMPIC_WGRAD_SHARE=$m timeout -k 10 300 python scripts/bench_train.py --steps 5 --warmup 2 > gpurun_out/r4_sh${m}_$i.log 2>&1
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4_sh*_*.log')):
    l=[l for l in open(f) if l.startswith('{')]
    if l:
        d=json.loads(l[-1]); print(f, d['ms_per_step'], d['phase_ms'], d['config']['loss'])
    else: print(f, 'no result')
PY
