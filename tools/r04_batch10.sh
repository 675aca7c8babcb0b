#!/bin/bash
# round 4: full GPU suite on the build with the row-range census, the tiled layout conversion, the batched insert and the
# tuner's several-passes-per-launch decision; then the A/Bs and the rates that changed
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04/tests10.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r04/tests10.log
grep -q " passed" gpurun_out/r04/tests10.log || exit 1
echo "== region grid: row-range census on / off"
CCP_GS_MASK_ROWS=1 timeout -k 10 300 python tools/fused_ab.py region > gpurun_out/r04/region_rows_on.jsonl 2>&1
CCP_GS_MASK_ROWS=0 timeout -k 10 300 python tools/fused_ab.py region > gpurun_out/r04/region_rows_off.jsonl 2>&1
grep -h "^{" gpurun_out/r04/region_rows_on.jsonl gpurun_out/r04/region_rows_off.jsonl | cut -c1-250
echo "== 4096^2 x 3: tuned (several passes per launch decided by the tuner), debug lines"
CCP_GS_DEBUG=1 timeout -k 10 300 python tools/fused_ab.py mid 2>&1 | grep -E "tune: 8 passes|tuned" | cut -c1-300 | tee gpurun_out/r04/mid_tuned.txt
echo "== lab3 modify scenario"
timeout -k 10 300 python tools/insert_bench.py --no-mask > gpurun_out/r04/insert_bench.json 2> gpurun_out/r04/insert_bench.err; tail -c 900 gpurun_out/r04/insert_bench.json
echo "== reference-order rates"
timeout -k 10 300 python tools/lex_grid_bench.py 2>&1 | grep "^{" | tee gpurun_out/r04/lex_final2.jsonl | cut -c1-330
