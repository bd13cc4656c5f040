#!/bin/bash
# Everything profiles/<tag>_* is made of, in ONE gpurun call (about 6 minutes of box time):
#
#   gpurun --timeout 1200 -- 'bash scripts/profile_round.sh r04'
#
# 2. rocprofv3 --kernel-trace --stats of the bench command                       -> <tag>_kernel_stats.csv, <tag>_summary.md
# 3. the same trace read as a timeline (two-branch graph overlap)                -> <tag>_timeline.md
# 4. --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (eager launches)    -> <tag>_pmc_traffic.json (+ table in summary)
# 5. SQ / GRBM counter pass                                                      -> <tag>_mfma_util.md
# 6. bench.py (default run, with cpu_baseline) and bench.py --dtype bf16           -> <tag>_bench.json, <tag>_bench_bf16.json
# 7. scripts/bench_train.py under --kernel-trace --stats and under the SQ counters  -> <tag>_first_train_kernel_stats_top40.csv, <tag>_first_train_mfma_util.md
# Counter passes never share a run with a trace domain other than --kernel-trace (gpurun refuses that).
set -e -o pipefail
tag=${1:-rXX}
export TMPDIR=/tmp
root=$(pwd)
out=$root/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out" "$root/gpurun_out/profiles"
B="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-bf16 --no-train"      # (the secondary legs of the default run would mix their steps into the trace)

echo "[2] kernel trace + stats"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o tr -- python3 $B > "$out/trace.log" 2>&1
st=$(find "$out/trace" -name 'tr_kernel_stats.csv' | head -1); tr=$(find "$out/trace" -name 'tr_kernel_trace.csv' | head -1)
cp "$st" "$root/profiles/${tag}_kernel_stats.csv"
echo "[3] timeline"; python3 scripts/trace_overlap.py "$tag" "$tr" > "$out/timeline.log"

echo "[4] traffic counters"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/fetch" -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bf16 --no-train --no-graph > "$out/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/write" -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bf16 --no-train --no-graph > "$out/write.log" 2>&1
fc=$(find "$out/fetch" -name 'f_counter_collection.csv' | head -1); wc=$(find "$out/write" -name 'w_counter_collection.csv' | head -1)
# bench.py --no-graph runs warmup + steps + 3 (plan build, first eager run, HIP-event pass) forward passes: count them from the trace
passes=$(grep -c s2d_input_kernel "$(find "$out/fetch" -name 'f_kernel_trace.csv' | head -1)")
python3 scripts/pmc_summary.py "$tag" "$st" "$fc" "$wc" "$passes" > "$out/pmc_summary.log"
{ echo "# fabric traffic per kernel configuration ($tag)"; echo; echo "Separate \`--pmc FETCH_SIZE\` / \`--pmc WRITE_SIZE\` passes of \`bench.py --steps 2 --warmup 1 --no-graph\` (scripts/pmc_per_kernel.py; reads = 2 x FETCH_SIZE, the gfx950 correction; grid in threads)."; echo;
  python3 scripts/pmc_per_kernel.py "$fc" "$wc" "$(find "$out/fetch" -name 'f_kernel_trace.csv' | head -1)"; } > "$root/profiles/${tag}_traffic_per_kernel.md"

echo "[5] MFMA utilisation counters"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --kernel-trace --output-format csv -d "$out/sq" -o s -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bf16 --no-train --no-graph > "$out/sq.log" 2>&1
python3 scripts/pmc_mfma.py "$(find "$out/sq" -name 's_counter_collection.csv' | head -1)" "$(find "$out/sq" -name 's_kernel_trace.csv' | head -1)" > "$root/profiles/${tag}_mfma_util.md"

echo "[6] bench (after the counter passes: its traffic field reads profiles/${tag}_pmc_traffic.json)"
python3 bench.py > "$out/bench.json" 2> "$out/bench.err"; tail -1 "$out/bench.json" > "$root/profiles/${tag}_bench.json"
python3 bench.py --dtype bf16 --no-cpu-baseline --no-bf16 --no-train > "$out/bench_bf16.json" 2>> "$out/bench.err"; tail -1 "$out/bench_bf16.json" > "$root/profiles/${tag}_bench_bf16.json"
echo "[7] first-stage training step: kernel stats (hipGraph replay) and MFMA counters (eager launches)"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/ttrace" -o tt -- python3 scripts/bench_train.py --steps 3 --warmup 2 > "$out/ttrace.log" 2>&1
tst=$(find "$out/ttrace" -name 'tt_kernel_stats.csv' | head -1)
head -41 "$tst" > "$root/profiles/${tag}_first_train_kernel_stats_top40.csv"
tail -1 "$out/ttrace.log" > "$root/profiles/${tag}_first_train_bench_under_rocprof.json"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --kernel-trace --output-format csv -d "$out/tsq" -o ts -- python3 scripts/bench_train.py --steps 1 --warmup 1 --no-graph > "$out/tsq.log" 2>&1
python3 scripts/pmc_mfma.py "$(find "$out/tsq" -name 'ts_counter_collection.csv' | head -1)" "$(find "$out/tsq" -name 'ts_kernel_trace.csv' | head -1)" > "$root/profiles/${tag}_first_train_mfma_util.md"
rm -rf "$out/ttrace" "$out/tsq"
cp "$root"/profiles/${tag}_* "$root/gpurun_out/profiles/"
rm -rf "$out/trace" "$out/fetch" "$out/write" "$out/sq"          # raw traces stay on the box (tens of MB)
echo "[done]"; cat "$root/profiles/${tag}_bench.json"
