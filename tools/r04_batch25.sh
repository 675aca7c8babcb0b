#!/bin/bash
# round 4: layout conversions with paired anti-diagonals — parity, then kernel times under rocprofv3
mkdir -p gpurun_out/r04
timeout -k 10 240 python -m pytest tests/test_gpu_lex.py -m gpu -x -q > gpurun_out/r04/lex_tests_b25.log 2>&1
echo "lex tests rc=$?"; tail -2 gpurun_out/r04/lex_tests_b25.log
grep -q " passed" gpurun_out/r04/lex_tests_b25.log || exit 1
grep -q "failed" gpurun_out/r04/lex_tests_b25.log && exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof25
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof25 -o lex -- python3 $GRAFT_REPO_ROOT/tools/lex_grid_bench.py > $GRAFT_REPO_ROOT/gpurun_out/r04/lex_b25.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof25 -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r04/kernel_stats_lex_b25.csv
grep -h "lex" gpurun_out/r04/kernel_stats_lex_b25.csv | cut -c1-200
grep "^{" gpurun_out/r04/lex_b25.log | cut -c1-200
