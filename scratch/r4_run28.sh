#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python bench.py > gpurun_out/r4_bench_final.json 2> gpurun_out/r4_bench_final.err; echo "bench rc=$?"; tail -1 gpurun_out/r4_bench_final.json | cut -c1-260
timeout -k 10 150 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4_smoke.log 2>&1; tail -1 gpurun_out/r4_smoke.log
timeout -k 10 500 python -m pytest tests/test_gpu_model.py tests/test_gpu_golden.py tests/test_gpu_train.py -q > gpurun_out/r4_t28.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4_t28.log
