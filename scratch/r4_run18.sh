#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/tt
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tt -o tt -- python3 scripts/bench_train.py --steps 3 --warmup 2 > gpurun_out/r4_tt.log 2>&1; echo "trace rc=$?"
tail -1 gpurun_out/r4_tt.log | cut -c1-400
tr=$(find gpurun_out/tt -name 'tt_kernel_trace.csv' | head -1)
python3 scripts/train_timeline.py "$tr" > gpurun_out/r4_train_timeline.md; cat gpurun_out/r4_train_timeline.md
rm -rf gpurun_out/tt
