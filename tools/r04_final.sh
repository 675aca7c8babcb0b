#!/bin/bash
# round 4, final tree: the whole GPU suite, the reference-order rates, the default bench
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04/tests_final.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r04/tests_final.log
grep -q " passed" gpurun_out/r04/tests_final.log || exit 1
timeout -k 10 300 python tools/lex_grid_bench.py 2>&1 | grep "^{" | tee gpurun_out/r04/lex_final_tree.jsonl | cut -c1-330
rm -f gpurun_out/cpu_baseline_phases.log
( time timeout -k 10 900 python bench.py ) > gpurun_out/r04/bench_final.json 2> gpurun_out/r04/bench_final.err
echo "bench rc=$?"; tail -3 gpurun_out/r04/bench_final.err
