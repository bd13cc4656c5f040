#!/bin/bash
# round 4, graph-destroy probes: stand-alone ingredient masks, then the library-level sequence with
# the destruction switched on, under rocgdb for a native backtrace
mkdir -p gpurun_out
( for m in 0 1 3 7 15 31 63; do timeout -k 5 120 scratch/probe/graph_destroy $m 24 4; echo "exit $?"; done ) > gpurun_out/r4_gd_standalone.log 2>&1
tail -4 gpurun_out/r4_gd_standalone.log
GDB="rocgdb -batch -ex 'handle SIG33 nostop noprint' -ex 'handle SIG34 nostop noprint' -ex 'handle SIGUSR1 nostop noprint' -ex run -ex bt -ex 'info registers rip' -ex 'thread apply all bt 8' --args"
VAMPIC_GRAPH_DESTROY=1 timeout -k 10 420 bash -c "$GDB python scratch/probe/graph_destroy_lib.py 3" > gpurun_out/r4_gd_lib_destroy.log 2>&1
echo "lib destroy rc=$?"; grep -n "SIGSEGV\|Segmentation\|done: no fault\|round" gpurun_out/r4_gd_lib_destroy.log | head -12
VAMPIC_GRAPH_DESTROY=1 VAMPIC_GRAPH_KEEP_TEMPLATE=1 timeout -k 10 300 python scratch/probe/graph_destroy_lib.py 3 > gpurun_out/r4_gd_lib_keep.log 2>&1
echo "lib destroy+keep rc=$?"; tail -2 gpurun_out/r4_gd_lib_keep.log
if ! grep -q "SIGSEGV\|Segmentation" gpurun_out/r4_gd_lib_destroy.log; then
  # the short sequence did not fault: the order that did (once, under the debugger; not a loop)
  VAMPIC_GRAPH_DESTROY=1 timeout -k 10 700 bash -c "$GDB python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_golden.py tests/test_gpu_config_variants.py tests/test_gpu_bitstream.py -x -q" > gpurun_out/r4_gd_suite_destroy.log 2>&1
  echo "suite order rc=$?"; grep -n "SIGSEGV\|Segmentation\|passed\|failed" gpurun_out/r4_gd_suite_destroy.log | head
fi
