#!/bin/bash
# A/B on the same box: A = scratch/libvampic_A.so (reference build), B = in-tree library; interleaved runs
cd /root/repo
for i in 1 2; do
  VAMPIC_LIB=/root/repo/scratch/libvampic_A.so timeout -k 10 200 python scratch/conv_bench.py > gpurun_out/ab_A$i.log 2>&1
  timeout -k 10 200 python scratch/conv_bench.py > gpurun_out/ab_B$i.log 2>&1
done
paste gpurun_out/ab_A1.log gpurun_out/ab_B1.log gpurun_out/ab_A2.log gpurun_out/ab_B2.log | awk -F'\t' '{split($1,a," "); split($2,b," "); split($3,c," "); split($4,d," "); print a[1], a[2], a[3], a[4], " A:", a[7], c[7], " B:", b[7], d[7]}' | tail -14
for i in 1 2; do
VAMPIC_LIB=/root/repo/scratch/libvampic_A.so timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/A /'
timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/B /'
done
