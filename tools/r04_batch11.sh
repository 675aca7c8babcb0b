#!/bin/bash
# round 4: full GPU suite (pass-through waves, wavefront fences in k_fused_multi, batched insert test), reference-order rates,
# the tuner's several-passes-per-launch decision at 4096^2 x 3, then the default bench
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04/tests11.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r04/tests11.log
grep -q " passed" gpurun_out/r04/tests11.log || exit 1
timeout -k 10 300 python tools/lex_grid_bench.py 2>&1 | grep "^{" | tee gpurun_out/r04/lex_final3.jsonl | cut -c1-330
CCP_GS_DEBUG=1 timeout -k 10 300 python tools/fused_ab.py mid 2>&1 | grep -E "tune: 8 passes|tuned" | cut -c1-300 | tee gpurun_out/r04/mid_tuned2.txt
rm -f gpurun_out/cpu_baseline_phases.log
( time timeout -k 10 900 python bench.py ) > gpurun_out/r04/bench_default4.json 2> gpurun_out/r04/bench_default4.err
echo "bench rc=$?"; tail -3 gpurun_out/r04/bench_default4.err
