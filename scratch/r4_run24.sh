#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_collectives.py -q -x -s -k two_rank --durations=3 > gpurun_out/r4_t24.log 2>&1; echo "tests rc=$?"; grep -a "two-rank exchange\|passed\|failed\|call " gpurun_out/r4_t24.log | cut -c1-600
