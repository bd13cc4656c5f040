#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=40 > gpurun_out/r4_gputests3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_gputests3.log
tail -60 gpurun_out/r4_gputests3.log
