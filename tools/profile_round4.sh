#!/bin/bash
# rocprofv3 passes of round 4 (run on the GPU box from the repo root): kernel trace + stats of the default bench.py
# command; FETCH_SIZE / WRITE_SIZE passes (separate runs, as the guide prescribes) of the bench's timed kernels and of
# the configs[1] / [4] / reference-order / CG kernels; summaries under gpurun_out/r04/prof/, copied to profiles/ by hand.
set -o pipefail
out=$PWD/gpurun_out/r04/prof
mkdir -p "$out"
export TMPDIR=/tmp
lean="--no-configs --no-cpu-baseline --no-converge --no-parity --no-reference-order"
what=${1:-all}
if [ "$what" = kt ] || [ "$what" = all ]; then
    t0=$(date +%s)
    python3 bench.py > "$out/bench_n1.json" 2> "$out/bench_n1.err" &
    pid=$!
    while kill -0 $pid 2>/dev/null; do echo "bench.py running ($(( $(date +%s) - t0 )) s)"; sleep 20; done
    wait $pid; echo "bench rc=$? after $(( $(date +%s) - t0 )) s"; echo "$(( $(date +%s) - t0 ))" > "$out/bench_n1.seconds"
    rocprofv3 --output-format csv --kernel-trace --stats -d "$out/kt_bench" -o bench -- python3 bench.py --no-cpu-baseline > "$out/bench_under_kt.json" 2> "$out/kt_bench.log" || exit 1
    cp "$(find "$out/kt_bench" -name '*kernel_stats.csv' | head -1)" "$out/kernel_stats_bench.csv"
    echo "kt bench done"
    # the timed passes alone (no configs, no convergence run): the dominant kernel's average is the 16384^2 pass
    rocprofv3 --output-format csv --kernel-trace --stats -d "$out/kt_lean" -o bench -- python3 bench.py $lean > "$out/bench_lean_under_kt.json" 2> "$out/kt_lean.log" || exit 1
    cp "$(find "$out/kt_lean" -name '*kernel_stats.csv' | head -1)" "$out/kernel_stats_bench_timed_passes_only.csv"
    echo "kt lean done"
    [ "$what" = kt ] && { find "$out" -name '*kernel_trace.csv' -size +5M -delete; exit 0; }
fi
pass() {  # tag, command...
    tag=$1; shift
    for c in FETCH_SIZE WRITE_SIZE; do
        timeout -k 10 400 rocprofv3 --output-format csv --pmc $c -d "$out/pmc_${c}_$tag" -o k -- "$@" > "$out/pmc_${c}_$tag.txt" 2> "$out/pmc_${c}_$tag.log" \
            && python3 tools/pmc_summary.py "$(find "$out/pmc_${c}_$tag" -name '*counter_collection.csv' | head -1)" "$out/pmc_${c}_$tag.csv"
        echo "pmc $c $tag rc=$?"
    done
}
pass bench python3 bench.py $lean --steps 3 --warmup 1
pass mid python3 tools/profile_kernels.py mid
pass region python3 tools/profile_kernels.py region
pass sell python3 tools/profile_kernels.py gs
pass lex python3 tools/profile_kernels.py --sweeps 64 lex
pass cg python3 tools/profile_kernels.py cg
# SQ counters of the reference-order sweep (how busy the SIMDs are over the launch)
timeout -k 10 400 rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY -d "$out/pmc_sq_lex" -o k -- python3 tools/profile_kernels.py --sweeps 64 lex > "$out/pmc_sq_lex.txt" 2> "$out/pmc_sq_lex.log" \
    && python3 tools/pmc_summary.py "$(find "$out/pmc_sq_lex" -name '*counter_collection.csv' | head -1)" "$out/pmc_sq_lex.csv"
echo "pmc sq lex rc=$?"
ROUND=r04 python3 tools/make_traffic.py "$out" "$out/traffic.json"
find "$out" -name '*counter_collection.csv' -delete
find "$out" -name '*kernel_trace.csv' -size +5M -delete
ls "$out"
