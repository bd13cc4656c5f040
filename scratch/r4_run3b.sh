#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 200 python scratch/wgrad_bench.py > gpurun_out/r4_wgbench_lds.log 2>&1; tail -25 gpurun_out/r4_wgbench_lds.log
VAMPIC_WGRAD_LDS=0 timeout -k 10 200 python scratch/wgrad_bench.py > gpurun_out/r4_wgbench_old.log 2>&1; tail -25 gpurun_out/r4_wgbench_old.log
timeout -k 10 200 python -m pytest tests/test_gpu_wgrad_lds.py tests/test_gpu_train.py -q -k "mixed or module_level" > gpurun_out/r4_t3.log 2>&1; tail -3 gpurun_out/r4_t3.log
bash scratch/r4_run3.sh
