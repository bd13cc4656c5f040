#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 120 python scratch/dbg_wg_mixed.py > gpurun_out/r4_dbg_mixed.log 2>&1; tail -20 gpurun_out/r4_dbg_mixed.log
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_gs.py tests/test_gpu_first_train.py tests/test_gpu_golden.py tests/test_gpu_ops.py -q > gpurun_out/r4_t2b.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t2b.log
tail -5 gpurun_out/r4_t2b.log
