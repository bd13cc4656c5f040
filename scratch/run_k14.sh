#!/bin/bash
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_train_gs.py tests/test_gpu_ops.py -q -x -s > gpurun_out/r2_k14_tests.log 2>&1
echo "k14+ops tests rc=$?"; tail -25 gpurun_out/r2_k14_tests.log | cut -c1-300
for r in 1 2; do
  VAMPIC_SPEC=0 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed 's/^/one-role /'
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed 's/^/auto     /'
done
