#!/bin/bash
# round 4, call 1: the GPU suite on the round's first build, then the default bench (CPU baseline on the headline system in a
# child process; its per-phase log goes to gpurun_out/cpu_baseline_phases.log)
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04/tests1.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r04/tests1.log
tail -5 gpurun_out/r04/tests1.log
nproc; free -g | head -2
( time timeout -k 10 900 python bench.py ) > gpurun_out/r04/bench_default.json 2> gpurun_out/r04/bench_default.err
echo "bench rc=$?"
cat gpurun_out/cpu_baseline_phases.log
tail -3 gpurun_out/r04/bench_default.err
