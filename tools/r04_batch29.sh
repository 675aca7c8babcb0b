#!/bin/bash
# round 4: PMC passes of the reference-order sweep again (the kernel changed: shifted strips, double-buffered edge
# values); the other kernels' summaries are taken from profiles/ as committed, traffic.json is rebuilt from all of them
set -o pipefail
out=$PWD/gpurun_out/r04/prof
mkdir -p "$out"
export TMPDIR=/tmp
for f in profiles/r04_pmc_*.csv; do cp "$f" "$out/$(basename "$f" | sed 's/^r04_//')"; done
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --output-format csv --pmc $c -d "$out/pmc_${c}_lex" -o k -- python3 tools/profile_kernels.py --sweeps 64 lex > "$out/pmc_${c}_lex.txt" 2> "$out/pmc_${c}_lex.log" \
        && python3 tools/pmc_summary.py "$(find "$out/pmc_${c}_lex" -name '*counter_collection.csv' | head -1)" "$out/pmc_${c}_lex.csv"
    echo "pmc $c lex rc=$?"
done
timeout -k 10 400 rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY -d "$out/pmc_sq_lex" -o k -- python3 tools/profile_kernels.py --sweeps 64 lex > "$out/pmc_sq_lex.txt" 2> "$out/pmc_sq_lex.log" \
    && python3 tools/pmc_summary.py "$(find "$out/pmc_sq_lex" -name '*counter_collection.csv' | head -1)" "$out/pmc_sq_lex.csv"
echo "pmc sq lex rc=$?"
ROUND=r04 python3 tools/make_traffic.py "$out" "$out/traffic.json"
find "$out" -name '*counter_collection.csv' -delete
grep -h "k_lex_wg" "$out"/pmc_*_lex.csv | cut -c1-300
