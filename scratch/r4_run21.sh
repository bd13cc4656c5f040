#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_first_train.py -q -x --durations=12 > gpurun_out/r4_t21.log 2>&1; echo "tests rc=$?"; tail -24 gpurun_out/r4_t21.log
