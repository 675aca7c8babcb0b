#!/bin/bash
# round 4: the strips' head / tail bodies — the new tests, then the whole GPU suite, then the reference-order rates
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_lex.py -m gpu -x -q > gpurun_out/r04/lex_tests_b20.log 2>&1
echo "lex tests rc=$?"; tail -3 gpurun_out/r04/lex_tests_b20.log
grep -q " passed" gpurun_out/r04/lex_tests_b20.log || exit 1
grep -q "failed" gpurun_out/r04/lex_tests_b20.log && exit 1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04/tests_b20.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r04/tests_b20.log
timeout -k 10 300 python tools/lex_grid_bench.py 2>&1 | grep "^{" | tee gpurun_out/r04/lex_b20.jsonl | cut -c1-200
