#!/bin/bash
mkdir -p gpurun_out
bash scripts/profile_round.sh r04 > gpurun_out/r4_profile_round.log 2>&1; rc=$?; echo "profile rc=$rc"; tail -3 gpurun_out/r4_profile_round.log | cut -c1-600
[ $rc -eq 0 ] || exit $rc
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tt -o tt -- python3 scripts/bench_train.py --steps 3 --warmup 2 > gpurun_out/r4_tt.log 2>&1 && \
  python3 scripts/train_timeline.py "$(find gpurun_out/tt -name 'tt_kernel_trace.csv' | head -1)" > gpurun_out/profiles/r04_first_train_timeline.md
rm -rf gpurun_out/tt
VAMPIC_DIST_BACKEND=gloo timeout -k 10 420 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-bf16 > gpurun_out/r4_gloo2.log 2>&1; echo "gloo 2-rank rc=$?"; tail -1 gpurun_out/r4_gloo2.log | cut -c1-300
