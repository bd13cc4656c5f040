#!/bin/bash
# per-kernel-configuration fabric traffic (scripts/pmc_per_kernel.py)
set -e -o pipefail
export TMPDIR=/tmp
out=$(pwd)/gpurun_out/pmc_pk; rm -rf $out; mkdir -p $out
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph > $out/write.log 2>&1
python3 scripts/pmc_per_kernel.py $(find $out/fetch -name f_counter_collection.csv) $(find $out/write -name w_counter_collection.csv) $(find $out/fetch -name f_kernel_trace.csv) > $out/per_kernel.md
rm -rf $out/fetch $out/write
