#!/bin/bash
# round 4: reference-order tests on the final persistent build, the reference-order rates, then the default bench
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_lex.py tests/test_gpu_region.py -m gpu -x -q > gpurun_out/r04/tests9.log 2>&1
echo "tests rc=$?"; tail -2 gpurun_out/r04/tests9.log
grep -q " passed" gpurun_out/r04/tests9.log || exit 1
timeout -k 10 300 python tools/lex_grid_bench.py > gpurun_out/r04/lex_final.jsonl 2>&1
grep -v amdgpu.ids gpurun_out/r04/lex_final.jsonl | cut -c1-330
rm -f gpurun_out/cpu_baseline_phases.log
( time timeout -k 10 900 python bench.py ) > gpurun_out/r04/bench_default3.json 2> gpurun_out/r04/bench_default3.err
echo "bench rc=$?"; tail -3 gpurun_out/r04/bench_default3.err
