#!/bin/bash
# round 4, call 2: the GPU suite after the prune and the ADVICE fixes, the default bench (CPU baseline: the 16384^2 system through
# the reference's 64-bit IndexType), per-workgroup traces of the reference-order sweep
set -o pipefail
mkdir -p gpurun_out/r04
rm -f gpurun_out/cpu_baseline_phases.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04/tests2.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r04/tests2.log
( time timeout -k 10 900 python bench.py ) > gpurun_out/r04/bench_default2.json 2> gpurun_out/r04/bench_default2.err
echo "bench rc=$?"; cat gpurun_out/cpu_baseline_phases.log; tail -4 gpurun_out/r04/bench_default2.err
for cfg in "16384 16384 128" "512 512 100" "4096 4096 64"; do
  set -- $cfg
  timeout -k 10 300 python tools/lex_trace.py run $1 $2 $3 gpurun_out/r04/lex_trace_$1.bin >> gpurun_out/r04/lex_trace.jsonl 2>&1 && \
  python tools/lex_trace.py show gpurun_out/r04/lex_trace_$1.bin >> gpurun_out/r04/lex_trace.jsonl 2>&1
  rm -f gpurun_out/r04/lex_trace_$1.bin
done
tail -c 6000 gpurun_out/r04/lex_trace.jsonl
