#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 100 scratch/probe/stack_tail2 32 2 200 > gpurun_out/r4_stack_tail2.log 2>&1; echo "prototype rc=$?"; cat gpurun_out/r4_stack_tail2.log
timeout -k 10 100 scratch/probe/stack_tail2 32 10 200 >> gpurun_out/r4_stack_tail2.log 2>&1; tail -1 gpurun_out/r4_stack_tail2.log
timeout -k 10 200 python scratch/conv_bench.py > gpurun_out/r4_convbench.log 2>&1; grep "128->64\|64->32" gpurun_out/r4_convbench.log
timeout -k 10 700 python -m pytest tests/test_gpu_first_train.py tests/test_gpu_config_variants.py -q -k "variants" > gpurun_out/r4_t13.log 2>&1; tail -5 gpurun_out/r4_t13.log
