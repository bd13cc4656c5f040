#!/bin/bash
# wgrad tile sweep: forced wave grids x chunk sizes on the k3 / k5 shapes of a first_train step
mkdir -p gpurun_out
for kp in 64 32; do
  for t in "4,2" "3,2" "2,2" "4,1" "3,3"; do
    echo "== tile $t kp $kp" >> gpurun_out/r4_wgsweep.log
    ONLY=3,5 VAMPIC_WGRAD_TILE=$t VAMPIC_WGRAD_KP=$kp timeout -k 10 150 python scratch/wgrad_bench.py >> gpurun_out/r4_wgsweep.log 2>&1 || echo "FAILED $t $kp" >> gpurun_out/r4_wgsweep.log
  done
done
echo "== automatic" >> gpurun_out/r4_wgsweep.log
timeout -k 10 200 python scratch/wgrad_bench.py >> gpurun_out/r4_wgsweep.log 2>&1
timeout -k 10 300 python -m pytest tests/test_gpu_wgrad_lds.py -q > gpurun_out/r4_t4.log 2>&1; tail -3 gpurun_out/r4_t4.log
VAMPIC_WGRAD_KP=32 timeout -k 10 300 python -m pytest tests/test_gpu_wgrad_lds.py -q >> gpurun_out/r4_t4.log 2>&1; tail -3 gpurun_out/r4_t4.log
VAMPIC_WGRAD_TILE=3,2 timeout -k 10 300 python -m pytest tests/test_gpu_wgrad_lds.py -q >> gpurun_out/r4_t4.log 2>&1; tail -3 gpurun_out/r4_t4.log
# the formerly crashing test order once more with the destruction ON, WITHOUT the debugger (r4_run3: under rocgdb 84 passed),
# native backtrace on a fault through the preloaded handler
VAMPIC_GRAPH_DESTROY=1 LD_PRELOAD=$PWD/scratch/probe/segv_bt.so timeout -k 10 600 python -m pytest -p no:faulthandler tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_golden.py tests/test_gpu_config_variants.py tests/test_gpu_bitstream.py -x -q > gpurun_out/r4_gd_suite_nogdb.log 2>&1
echo "suite order, destroy on, no debugger: rc=$?"; tail -4 gpurun_out/r4_gd_suite_nogdb.log
