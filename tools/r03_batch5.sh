#!/bin/bash
# round 3, batch 5: masked multi-pass parity, CG loops, region grid multi A/B, timeline of the edges-first pass
OUT=gpurun_out/r03
mkdir -p $OUT
export TMPDIR=/tmp
echo "== multi parity (incl. mask grids)"
timeout -k 10 600 python -m pytest tests/test_gpu_multi.py tests/test_gpu_region.py -x -q -m gpu > $OUT/b5_tests.log 2>&1; rc=$?; echo "rc=$rc"; tail -12 $OUT/b5_tests.log
echo "== CG loops"
rm -f $OUT/b5_cg.jsonl
for f in 1 2 0; do
  CCP_GS_CG_FUSED=$f timeout -k 10 300 python tools/cg_bench.py --size 8192 --height 4096 --channels 3 --iters 50 >> $OUT/b5_cg.jsonl 2>> $OUT/b5_cg.err
  CCP_GS_CG_FUSED=$f timeout -k 10 300 python tools/cg_bench.py --size 16384 --channels 1 --iters 30 >> $OUT/b5_cg.jsonl 2>> $OUT/b5_cg.err
done
cat $OUT/b5_cg.jsonl
echo "== region grid, multi 0/1"
rm -f $OUT/b5_region.jsonl
for m in 0 1; do CCP_GS_MULTI=$m timeout -k 10 600 python tools/fused_ab.py region >> $OUT/b5_region.jsonl 2>> $OUT/b5_region.err; done
cat $OUT/b5_region.jsonl
echo "== timeline of a block's plain and edges-first pass"
rm -f $OUT/b5_pass_trace.jsonl
for a in "16384 16384 1 8 198 8192 2048 64 0" "16384 16384 1 8 198 8192 2048 64 1"; do
  timeout -k 10 300 python tools/pass_trace.py $a >> $OUT/b5_pass_trace.jsonl 2>> $OUT/b5_pass_trace.err || echo "trace $a failed"
done
cat $OUT/b5_pass_trace.jsonl | cut -c1-1500
