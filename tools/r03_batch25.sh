#!/bin/bash
# round 3, batch 25: PMC traffic of the region grid at depth 8 (separate FETCH_SIZE / WRITE_SIZE passes)
out=$PWD/gpurun_out/r03/prof
mkdir -p "$out"
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf "$out/pmc_${c}_region"
    timeout -k 10 400 rocprofv3 --output-format csv --pmc $c -d "$out/pmc_${c}_region" -o k -- python3 tools/profile_kernels.py region > "$out/pmc_${c}_region.txt" 2> "$out/pmc_${c}_region.log" \
        && python3 tools/pmc_summary.py "$(find "$out/pmc_${c}_region" -name '*counter_collection.csv' | head -1)" "$out/pmc_${c}_region.csv"
    echo "pmc $c region rc=$?"
    grep -E "^kernel|k_fused_sweep_masked" "$out/pmc_${c}_region.csv"
done
find "$out" -name '*counter_collection.csv' -delete
