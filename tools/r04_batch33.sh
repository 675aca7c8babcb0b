#!/bin/bash
# round 4: forward layout conversion on sheared tiles — parity (reference-order + region tests), then kernel stats and rates
mkdir -p gpurun_out/r04
timeout -k 10 300 python -m pytest tests/test_gpu_lex.py tests/test_gpu_region.py -m gpu -x -q > gpurun_out/r04/lex_tests_b33.log 2>&1
echo "lex tests rc=$?"; tail -2 gpurun_out/r04/lex_tests_b33.log
grep -q " passed" gpurun_out/r04/lex_tests_b33.log || exit 1
grep -q "failed" gpurun_out/r04/lex_tests_b33.log && exit 1
bash tools/r04_batch25.sh
