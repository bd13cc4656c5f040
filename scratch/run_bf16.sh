#!/bin/bash
cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py -q -x -s > gpurun_out/r2_bf16_tests.log 2>&1
echo "bf16 tests rc=$?"; tail -15 gpurun_out/r2_bf16_tests.log | cut -c1-300
timeout -k 10 300 python bench.py --dtype bf16 --no-cpu-baseline --steps 10 > gpurun_out/r2_bench_bf16_cfg1.log 2>gpurun_out/r2_bench_bf16_cfg1.err; tail -c 1800 gpurun_out/r2_bench_bf16_cfg1.log; tail -3 gpurun_out/r2_bench_bf16_cfg1.err
timeout -k 10 300 python bench.py --dtype bf16 --batch 8 --height 512 --width 768 --no-cpu-baseline --steps 10 > gpurun_out/r2_bench_bf16_cfg2.log 2>gpurun_out/r2_bench_bf16_cfg2.err; tail -c 1800 gpurun_out/r2_bench_bf16_cfg2.log; tail -3 gpurun_out/r2_bench_bf16_cfg2.err
timeout -k 10 300 python bench.py --batch 8 --height 512 --width 768 --no-cpu-baseline --steps 10 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed 's/^/fp32 8x512x768 /'
