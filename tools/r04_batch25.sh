#!/bin/bash
# round 4: layout conversions with paired anti-diagonals — parity, then kernel times under rocprofv3
mkdir -p gpurun_out/r04
true



cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof25
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof25 -o lex -- python3 $GRAFT_REPO_ROOT/tools/lex_grid_bench.py > $GRAFT_REPO_ROOT/gpurun_out/r04/lex_b25.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof25 -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r04/kernel_stats_lex_b25.csv
grep -h "lex" gpurun_out/r04/kernel_stats_lex_b25.csv | cut -c1-200
grep "^{" gpurun_out/r04/lex_b25.log | cut -c1-200
