#!/bin/bash
# A/B on one box: bf16x3-plane intermediates inside the residual units of the 64x64 feature maps (VAMPIC_P3_MAX_PIXELS)
cd /root/repo
for i in 1 2; do
  for v in 16384 131072; do
    VAMPIC_P3_MAX_PIXELS=$v timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/P3_MAX=$v /"
  done
done
