#!/bin/bash
# round 4: strips that move 2T columns left per group — parity (lex tests under a short timeout), trace, rates
mkdir -p gpurun_out/r04
timeout -k 10 240 python -m pytest tests/test_gpu_lex.py -m gpu -x -q > gpurun_out/r04/lex_tests_b21.log 2>&1
echo "lex tests rc=$?"; tail -3 gpurun_out/r04/lex_tests_b21.log
grep -q " passed" gpurun_out/r04/lex_tests_b21.log || exit 1
grep -q "failed" gpurun_out/r04/lex_tests_b21.log && exit 1
timeout -k 10 120 python tools/lex_trace.py run 512 512 100 gpurun_out/r04/trace_512_100.bin || exit 1
python tools/lex_trace.py show gpurun_out/r04/trace_512_100.bin | tail -1 | cut -c1-700
python tools/lex_trace.py table gpurun_out/r04/trace_512_100.bin > gpurun_out/r04/trace_512_100.txt; sed -n 1,7p gpurun_out/r04/trace_512_100.txt
timeout -k 10 300 python tools/lex_grid_bench.py 2>&1 | grep "^{" | tee gpurun_out/r04/lex_b21.jsonl | cut -c1-200
