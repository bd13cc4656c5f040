#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 100 scratch/probe/stack_tail2 32 2 200 > gpurun_out/r4_stack_tail2b.log 2>&1; echo "prototype rc=$?"; cat gpurun_out/r4_stack_tail2b.log
timeout -k 10 100 scratch/probe/stack_tail2 32 8 200 >> gpurun_out/r4_stack_tail2b.log 2>&1 && tail -2 gpurun_out/r4_stack_tail2b.log
