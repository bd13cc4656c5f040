#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/r4_gputests4.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_gputests4.log
tail -14 gpurun_out/r4_gputests4.log
timeout -k 10 150 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4_smoke.log 2>&1; tail -2 gpurun_out/r4_smoke.log
