#!/bin/bash
# rocprofv3 passes for the reference-order sweep (k_lex_wg), run on the GPU box from the repo root:
# kernel trace + stats of tools/lex_grid_bench.py (all sizes, fixed count and stop rule), FETCH_SIZE / WRITE_SIZE
# of one 64-sweep run of the 16384^2 grid (8 groups of 8 sweeps, one launch).  Summaries under gpurun_out/$1/.
set -o pipefail
tag=${1:-r02_lex_wg}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --stats -d "$out/kt" -o k -- python3 tools/lex_grid_bench.py > "$out/lex_grid_bench.jsonl" 2> "$out/kt.log" || exit 1
cp "$out"/kt/k_kernel_stats.csv "$out/kernel_stats_lex_grid_bench.csv"
echo "kt done"
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --output-format csv --pmc $c -d "$out/pmc_${c}" -o k -- python3 tools/profile_kernels.py --sweeps 64 lex > "$out/pmc_${c}.txt" 2> "$out/pmc_${c}.log" \
        && python3 tools/pmc_summary.py "$(find "$out/pmc_${c}" -name '*counter_collection.csv' | head -1)" "$out/pmc_${c}_lex_wg_16384.csv"
    echo "pmc $c done rc=$?"
done
find "$out" -name '*counter_collection.csv' -delete
find "$out" -name '*kernel_trace.csv' -size +5M -delete
