#!/bin/bash
# round 3, batch 10: k_lex_wg2 (two pixels per lane and step) — parity under every reference-order test, then speed
OUT=gpurun_out/r03
mkdir -p $OUT
export TMPDIR=/tmp
echo "== parity with CCP_GS_LEX_MODE=wg2"
CCP_GS_LEX_MODE=wg2 timeout -k 10 500 python -m pytest tests/test_gpu_lex.py -x -q -m gpu > $OUT/b10_tests_lex.log 2>&1; rc=$?; echo "rc=$rc"; tail -15 $OUT/b10_tests_lex.log
[ $rc -ne 0 ] && exit 1
CCP_GS_LEX_MODE=wg2 timeout -k 10 400 python -m pytest tests/test_gpu_region.py tests/test_gpu_fullsize.py -x -q -m gpu > $OUT/b10_tests_region.log 2>&1; rc=$?; echo "rc=$rc"; tail -8 $OUT/b10_tests_region.log
echo "== speed"
for m in wg wg2; do CCP_GS_LEX_MODE=$m timeout -k 10 400 python tools/lex_grid_bench.py > $OUT/b10_lex_$m.jsonl 2> $OUT/b10_lex_$m.err; echo "rc=$?"; echo "mode $m"; cat $OUT/b10_lex_$m.jsonl; done
