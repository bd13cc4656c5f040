#!/bin/bash
# interleaved A/Bs on ONE box: (1) inference step, library with / without the training-only epilogue features compiled in;
# (2) first_train step with / without the deferred, shape-grouped weight-gradient launches
mkdir -p gpurun_out
for i in 1 2; do
  timeout -k 10 200 python bench.py --no-train --no-bf16 --no-cpu-baseline --steps 20 > gpurun_out/r4_epi1_$i.log 2>&1
  VAMPIC_LIB=$PWD/scratch/ab/libvampic_noepi.so timeout -k 10 200 python bench.py --no-train --no-bf16 --no-cpu-baseline --steps 20 > gpurun_out/r4_epi0_$i.log 2>&1
done
for i in 1 2; do
  timeout -k 10 300 python scripts/bench_train.py --steps 5 --warmup 2 > gpurun_out/r4_defer1_$i.log 2>&1
  VAMPIC_WGRAD_DEFER=0 timeout -k 10 300 python scripts/bench_train.py --steps 5 --warmup 2 > gpurun_out/r4_defer0_$i.log 2>&1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4_epi*_*.log')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); r=d['roofline']
    print(f, d['ms_per_step'], r['frac'], r['g_a_g_s']['frac'], r['g_a_g_s']['ms_per_step'])
for f in sorted(glob.glob('gpurun_out/r4_defer*_*.log')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f, d['ms_per_step'], d['phase_ms'])
PY
