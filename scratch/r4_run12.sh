#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_wgrad_lds.py tests/test_gpu_first_train.py -q -k "wgrad or plane or mixed or variants or slice_stack or full_size" > gpurun_out/r4_t12.log 2>&1; tail -6 gpurun_out/r4_t12.log
ONLY=1 timeout -k 10 200 python scratch/wgrad_bench.py > gpurun_out/r4_wgbench4.log 2>&1; tail -9 gpurun_out/r4_wgbench4.log
timeout -k 10 300 python scripts/bench_train.py --steps 5 --warmup 2 > gpurun_out/r4_train6.log 2>&1; tail -1 gpurun_out/r4_train6.log | cut -c1-300
