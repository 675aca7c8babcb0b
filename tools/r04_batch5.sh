#!/bin/bash
# round 4, call 5: persistent k_lex_wg with the strip as a non-inlined function — parity, rates, trace
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_lex.py tests/test_gpu_region.py -m gpu -x -q > gpurun_out/r04/tests5.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r04/tests5.log
grep -q " passed" gpurun_out/r04/tests5.log || exit 1
timeout -k 10 300 python tools/lex_grid_bench.py > gpurun_out/r04/lex_noinline.jsonl 2>&1
timeout -k 10 300 python tools/lex_trace.py run 16384 16384 128 gpurun_out/r04/lex_trace_ni.bin >> gpurun_out/r04/lex_noinline.jsonl 2>&1 && \
python tools/lex_trace.py show gpurun_out/r04/lex_trace_ni.bin >> gpurun_out/r04/lex_noinline.jsonl 2>&1
cat gpurun_out/r04/lex_noinline.jsonl | cut -c1-1300
