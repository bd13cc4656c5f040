#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/r4_stack_tail_swz.log
for v in 0 1 2; do echo "SWZ=$v" >> gpurun_out/r4_stack_tail_swz.log; timeout -k 10 60 scratch/probe/stack_tail3_s$v 32 2 200 >> gpurun_out/r4_stack_tail_swz.log 2>&1; echo "swz $v rc=$?"; done
cat gpurun_out/r4_stack_tail_swz.log
