#!/bin/bash
# round 4: the general block without its division sequence — parity first, then the 512^2 trace and the rates
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_lex.py -m gpu -x -q > gpurun_out/r04/lex_tests_b16.log 2>&1
echo "lex tests rc=$?"; tail -2 gpurun_out/r04/lex_tests_b16.log
grep -q " passed" gpurun_out/r04/lex_tests_b16.log || exit 1
for shape in "512 512 100" "16384 16384 128"; do
  set -- $shape
  timeout -k 10 120 python tools/lex_trace.py run $1 $2 $3 gpurun_out/r04/trace_$1_$3.bin || exit 1
  python tools/lex_trace.py show gpurun_out/r04/trace_$1_$3.bin | tail -1 | cut -c1-900
done
python tools/lex_trace.py table gpurun_out/r04/trace_512_100.bin | head -5
timeout -k 10 300 python tools/lex_grid_bench.py 2>&1 | grep "^{" | tee gpurun_out/r04/lex_b16.jsonl | cut -c1-200
