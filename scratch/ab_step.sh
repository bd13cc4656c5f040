#!/bin/bash
# whole-step A/B on one box: A = scratch/libvampic_A.so, B = in-tree library; interleaved, fp32 headline and the bf16 leg
cd /root/repo
for i in 1 2 3; do
VAMPIC_LIB=/root/repo/scratch/libvampic_A.so timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('A', d['ms_per_step'], d['bf16']['ms_per_step'])"
timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('B', d['ms_per_step'], d['bf16']['ms_per_step'])"
done
