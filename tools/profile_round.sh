#!/bin/bash
# rocprofv3 passes of one round (run on the GPU box from the repo root): kernel trace + stats of the default
# bench.py command, FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, as the guide prescribes) of the
# bench's timed kernels and of the configs[0]/[4] kernels.  Raw output under gpurun_out/$1/, per-kernel
# summaries (tools/pmc_summary.py) next to it; copy what is to be judged into profiles/.
set -o pipefail
tag=${1:-r02}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
lean="--no-configs --no-cpu-baseline --no-converge --no-parity --no-reference-order"
rocprofv3 --kernel-trace --stats -d "$out/kt_bench" -o bench -- python3 bench.py > "$out/bench_under_kt.json" 2> "$out/kt_bench.log" || exit 1
echo "kt bench done"
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d "$out/pmc_${c}_bench" -o bench -- python3 bench.py $lean > "$out/bench_under_$c.json" 2> "$out/pmc_${c}_bench.log" || exit 1
    python3 tools/pmc_summary.py "$(find "$out/pmc_${c}_bench" -name '*counter_collection.csv' | head -1)" "$out/pmc_${c}_bench.csv"
    echo "pmc $c bench done"
done
rocprofv3 --kernel-trace --stats -d "$out/kt_kernels" -o k -- python3 tools/profile_kernels.py > "$out/kt_kernels.txt" 2> "$out/kt_kernels.log" || exit 1
echo "kt kernels done"
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d "$out/pmc_${c}_kernels" -o k -- python3 tools/profile_kernels.py > /dev/null 2> "$out/pmc_${c}_kernels.log" || exit 1
    python3 tools/pmc_summary.py "$(find "$out/pmc_${c}_kernels" -name '*counter_collection.csv' | head -1)" "$out/pmc_${c}_kernels.csv"
    echo "pmc $c kernels done"
done
find "$out" -name '*kernel_stats.csv' -exec sh -c 'cp "$1" "$2/$(basename $(dirname $(dirname "$1")))_kernel_stats.csv"' _ {} "$out" \;
# the raw per-dispatch files are large: keep the summaries only
find "$out" -name '*counter_collection.csv' -delete
find "$out" -name '*kernel_trace.csv' -size +20M -delete
ls -la "$out"
