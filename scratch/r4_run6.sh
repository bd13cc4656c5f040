#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_wgrad_lds.py -q > gpurun_out/r4_t6.log 2>&1; tail -3 gpurun_out/r4_t6.log
timeout -k 10 200 python scratch/wgrad_bench.py > gpurun_out/r4_wgbench3.log 2>&1; tail -24 gpurun_out/r4_wgbench3.log
timeout -k 10 300 python scripts/bench_train.py --steps 5 --warmup 2 > gpurun_out/r4_train4.log 2>&1; tail -1 gpurun_out/r4_train4.log | cut -c1-400
timeout -k 10 400 python scratch/prof_first_train.py > gpurun_out/r4_prof3.log 2>&1
timeout -k 10 700 python -m pytest tests/test_gpu_first_train.py tests/test_gpu_train.py tests/test_gpu_train_gs.py tests/test_gpu_collectives.py -q > gpurun_out/r4_t6b.log 2>&1; tail -12 gpurun_out/r4_t6b.log
