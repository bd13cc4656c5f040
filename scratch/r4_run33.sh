#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 60 scratch/probe/stack_tail2 32 2 200 > gpurun_out/r4_stack_tail_tr.log 2>&1; echo "TR=4 rc=$?"
timeout -k 10 60 scratch/probe/stack_tail2_tr2 32 2 200 >> gpurun_out/r4_stack_tail_tr.log 2>&1; echo "TR=2 rc=$?"
timeout -k 10 60 scratch/probe/stack_tail2_tr2 32 8 200 >> gpurun_out/r4_stack_tail_tr.log 2>&1; echo "TR=2 x8 rc=$?"
cat gpurun_out/r4_stack_tail_tr.log
