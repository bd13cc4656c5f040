#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_train_gs.py -q -x -k "axpy or elementwise" > gpurun_out/r4_t29.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r4_t29.log
