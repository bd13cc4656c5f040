#!/bin/bash
# round 4, second GPU call: LDS-tiled weight gradients + GELU fused into the training epilogues
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_wgrad_lds.py tests/test_gpu_train.py tests/test_gpu_train_gs.py tests/test_gpu_first_train.py tests/test_gpu_ops.py -x -q > gpurun_out/r4_t2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_t2.log
tail -5 gpurun_out/r4_t2.log
timeout -k 10 300 python scripts/bench_train.py --steps 5 --warmup 2 > gpurun_out/r4_train2.log 2>&1
echo "train rc=$?"; tail -1 gpurun_out/r4_train2.log
VAMPIC_WGRAD_LDS=0 timeout -k 10 300 python scripts/bench_train.py --steps 5 --warmup 2 > gpurun_out/r4_train2_old.log 2>&1
echo "train(old wgrad) rc=$?"; tail -1 gpurun_out/r4_train2_old.log
timeout -k 10 400 python scratch/prof_first_train.py > gpurun_out/r4_prof2.log 2>&1
echo "prof rc=$?"
VAMPIC_FORCE_COLLECTIVES=1 timeout -k 10 300 python scripts/bench_finetune.py --steps 10 --warmup 3 > gpurun_out/r4_ft_forced2.log 2>&1
tail -1 gpurun_out/r4_ft_forced2.log
