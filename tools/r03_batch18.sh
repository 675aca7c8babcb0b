#!/bin/bash
# round 3, batch 18: k_lex_wg with the LDS reads of a step fetched during the step before (parity, then rates)
OUT=gpurun_out/r03
mkdir -p $OUT
echo "== parity"
timeout -k 10 900 python -m pytest tests/test_gpu_lex.py tests/test_gpu_region.py -x -q > $OUT/b18_tests.log 2>&1; rc=$?; tail -3 $OUT/b18_tests.log
[ $rc -ne 0 ] && { grep -n "^E \|Error\|FAILED" $OUT/b18_tests.log | head -20; exit 1; }
echo "== rates"
timeout -k 10 600 python tools/lex_grid_bench.py > $OUT/b18_lex.jsonl 2> $OUT/b18_lex.err; echo "rc=$?"; cat $OUT/b18_lex.jsonl; tail -3 $OUT/b18_lex.err
