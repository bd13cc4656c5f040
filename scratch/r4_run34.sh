#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 60 scratch/probe/stack_tail2 32 2 200 > gpurun_out/r4_stack_tail3.log 2>&1; echo "tail2 rc=$?"
timeout -k 10 60 scratch/probe/stack_tail3 32 2 200 >> gpurun_out/r4_stack_tail3.log 2>&1; echo "tail3 rc=$?"
timeout -k 10 60 scratch/probe/stack_tail3 32 8 200 >> gpurun_out/r4_stack_tail3.log 2>&1; echo "tail3 x8 rc=$?"
cat gpurun_out/r4_stack_tail3.log
