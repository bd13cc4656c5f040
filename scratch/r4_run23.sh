#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_collectives.py -q -x > gpurun_out/r4_t23.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/r4_t23.log
