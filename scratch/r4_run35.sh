#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -q -x -k "fused_stack_tail" > gpurun_out/r4_t35.log 2>&1; rc=$?; echo "op tests rc=$rc"; tail -5 gpurun_out/r4_t35.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -m pytest tests/test_gpu_model.py -q -x -k "fused_stack_tail or graph_replay" > gpurun_out/r4_t35b.log 2>&1; rc=$?; echo "model tests rc=$rc"; tail -5 gpurun_out/r4_t35b.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2; do
  for m in 0 1; do
    VAMPIC_STACK_TAIL=$m timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-bf16 --no-train > gpurun_out/r4_st${m}_$i.log 2>&1
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4_st*_*.log')):
    l=[l for l in open(f) if l.startswith('{')]
    if l:
        d=json.loads(l[-1]); r=d['roofline']; print(f, d['ms_per_step'], r['frac'], r['launches_per_step'], r['classes']['slice_chain'], r['classes']['lrp_prog'])
    else: print(f, 'no result')
PY
