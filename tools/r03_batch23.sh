#!/bin/bash
# round 3, batch 23: masked pass with the high-word factor window, depth up to 8
OUT=gpurun_out/r03
mkdir -p $OUT
echo "== parity"
timeout -k 10 900 python -m pytest tests/test_gpu_region.py tests/test_gpu_grid.py tests/test_gpu_multi.py tests/test_gpu_lex.py -x -q > $OUT/b23_tests.log 2>&1; rc=$?; tail -3 $OUT/b23_tests.log
[ $rc -ne 0 ] && { grep -n "^E \|Error\|FAILED" $OUT/b23_tests.log | head -20; exit 1; }
: > $OUT/b23_ab.jsonl
for rep in 1 2; do
  timeout -k 10 300 python tools/fused_ab.py region >> $OUT/b23_ab.jsonl 2>> $OUT/b23_ab.err || echo "fused_ab failed"
done
cat $OUT/b23_ab.jsonl
