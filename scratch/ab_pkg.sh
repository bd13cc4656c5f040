#!/bin/bash
# Python-level A/B on one box: A = scratch/pkg_A (package at HEAD), B variants = working tree with VAMPIC_HEADS=1|2
cd /root/repo
for i in 1 2; do
  VAMPIC_PKG_DIR=/root/repo/scratch/pkg_A timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/A(HEAD)   /'
  VAMPIC_HEADS=1 timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/B heads=1 /'
  VAMPIC_HEADS=2 timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/B heads=2 /'
done
