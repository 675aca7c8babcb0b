#!/bin/bash
# round 4: ticket order read from pinned host memory — parity, host laps, rates
mkdir -p gpurun_out/r04
timeout -k 10 240 python -m pytest tests/test_gpu_lex.py tests/test_gpu_region.py -m gpu -x -q > gpurun_out/r04/lex_tests_b28.log 2>&1
echo "lex tests rc=$?"; tail -2 gpurun_out/r04/lex_tests_b28.log
grep -q " passed" gpurun_out/r04/lex_tests_b28.log || exit 1
grep -q "failed" gpurun_out/r04/lex_tests_b28.log && exit 1
bash tools/r04_batch27.sh 2>&1 | grep -v "occupancy\|tickets sorted" | tail -24
timeout -k 10 300 python tools/lex_grid_bench.py 2>&1 | grep "^{" | tee gpurun_out/r04/lex_b28.jsonl | cut -c1-200
