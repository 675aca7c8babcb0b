#!/bin/bash
# round 3, batch 12: a row block of the mask grid of configs[4] (8 blocks of 1024 canvas rows), world-1 tests again
mkdir -p gpurun_out/r03
O=gpurun_out/r03
echo "== world-1 real RCCL tests"
timeout -k 10 300 python -m pytest tests/test_gpu_rccl.py -x -q > $O/b12_tests.log 2>&1; echo "rc=$?"; tail -3 $O/b12_tests.log
echo "== mask-grid row blocks"
: > $O/b12_mask_blocks.jsonl
for rank in 0 3 4 7; do
  for ghost in 32 64; do
    timeout -k 10 200 python tools/rank_block_bench.py 8 $ghost plain $rank mask >> $O/b12_mask_blocks.jsonl 2>> $O/b12_mask_blocks.err || echo "failed rank $rank ghost $ghost"
  done
done
cat $O/b12_mask_blocks.jsonl
