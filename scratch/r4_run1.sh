#!/bin/bash
# round 4, first GPU call: whole GPU suite (capture-allocation guard, forced collectives), training benches with and
# without the forced 1-rank RCCL group, bench.py with the new train legs
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputests1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_gputests1.log
tail -3 gpurun_out/r4_gputests1.log
timeout -k 10 300 python scripts/bench_train.py --steps 5 --warmup 2 > gpurun_out/r4_train_plain.log 2>&1 && \
VAMPIC_FORCE_COLLECTIVES=1 timeout -k 10 300 python scripts/bench_train.py --steps 5 --warmup 2 > gpurun_out/r4_train_forced.log 2>&1 && \
VAMPIC_FORCE_COLLECTIVES=1 timeout -k 10 300 python scripts/bench_finetune.py --steps 10 --warmup 3 > gpurun_out/r4_ft_forced.log 2>&1 && \
timeout -k 10 300 python scripts/bench_finetune.py --steps 10 --warmup 3 > gpurun_out/r4_ft_plain.log 2>&1 && \
VAMPIC_FORCE_COLLECTIVES=1 timeout -k 10 600 python bench.py > gpurun_out/r4_bench_forced.log 2>&1
echo "benches rc=$?"
tail -2 gpurun_out/r4_train_plain.log gpurun_out/r4_train_forced.log gpurun_out/r4_ft_forced.log gpurun_out/r4_ft_plain.log gpurun_out/r4_bench_forced.log
