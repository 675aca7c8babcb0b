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
rocprofv3 --output-format csv --kernel-trace --stats -d "$out/kt_bench" -o bench -- python3 bench.py --no-configs > "$out/bench_under_kt.json" 2> "$out/kt_bench.log" || exit 1
echo "kt bench done"
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --output-format csv --pmc $c -d "$out/pmc_${c}_bench" -o bench -- python3 bench.py $lean > "$out/bench_under_$c.json" 2> "$out/pmc_${c}_bench.log" || exit 1
    python3 tools/pmc_summary.py "$(find "$out/pmc_${c}_bench" -name '*counter_collection.csv' | head -1)" "$out/pmc_${c}_bench.csv"
    echo "pmc $c bench done"
done
rocprofv3 --output-format csv --kernel-trace --stats -d "$out/kt_kernels" -o k -- python3 tools/profile_kernels.py > "$out/kt_kernels.txt" 2> "$out/kt_kernels.log" || exit 1
echo "kt kernels done"
# PMC: the two bandwidth kernels at full size; the two launch-bound ones on smaller problems (tens of
# thousands of tiny dispatches, each serialised by the counter collection), each pass under its own time limit
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --output-format csv --pmc $c -d "$out/pmc_${c}_sell" -o k -- python3 tools/profile_kernels.py apply gs > "$out/pmc_${c}_sell.txt" 2> "$out/pmc_${c}_sell.log" \
        && python3 tools/pmc_summary.py "$(find "$out/pmc_${c}_sell" -name '*counter_collection.csv' | head -1)" "$out/pmc_${c}_sell.csv"
    echo "pmc $c sell done rc=$?"
    CCP_GS_LEX_SYNC_EVERY=32 timeout -k 10 300 rocprofv3 --output-format csv --pmc $c -d "$out/pmc_${c}_lex" -o k -- python3 tools/profile_kernels.py --grid 6144 --sweeps 4 lex > "$out/pmc_${c}_lex.txt" 2> "$out/pmc_${c}_lex.log" \
        && python3 tools/pmc_summary.py "$(find "$out/pmc_${c}_lex" -name '*counter_collection.csv' | head -1)" "$out/pmc_${c}_lex.csv"
    echo "pmc $c lex done rc=$?"
    timeout -k 10 300 rocprofv3 --output-format csv --pmc $c -d "$out/pmc_${c}_pipe" -o k -- python3 tools/profile_kernels.py --canvas 4096 pipe > "$out/pmc_${c}_pipe.txt" 2> "$out/pmc_${c}_pipe.log" \
        && python3 tools/pmc_summary.py "$(find "$out/pmc_${c}_pipe" -name '*counter_collection.csv' | head -1)" "$out/pmc_${c}_pipe.csv"
    echo "pmc $c pipe done rc=$?"
done
# the raw per-dispatch files are large: keep the summaries only
find "$out" -name '*counter_collection.csv' -delete
find "$out" -name '*kernel_trace.csv' -size +5M -delete
ls -la "$out"
