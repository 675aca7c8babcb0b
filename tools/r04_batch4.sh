#!/bin/bash
# round 4, call 4: persistent workgroups in k_lex_wg — reference-order parity tests, then rates and traces with a workgroup per strip (A/B)
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_lex.py tests/test_gpu_region.py -m gpu -x -q > gpurun_out/r04/tests4.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r04/tests4.log
grep -q " passed" gpurun_out/r04/tests4.log || exit 1
for p in 1 0; do
  echo "== CCP_GS_LEX_PERSISTENT=$p" >> gpurun_out/r04/lex_persistent.jsonl
  CCP_GS_LEX_PERSISTENT=$p timeout -k 10 300 python tools/lex_grid_bench.py >> gpurun_out/r04/lex_persistent.jsonl 2>&1
  CCP_GS_LEX_PERSISTENT=$p timeout -k 10 300 python tools/lex_trace.py run 16384 16384 128 gpurun_out/r04/lex_trace_p$p.bin >> gpurun_out/r04/lex_persistent.jsonl 2>&1 && \
  python tools/lex_trace.py show gpurun_out/r04/lex_trace_p$p.bin | tail -1 >> gpurun_out/r04/lex_persistent.jsonl 2>&1
done
cat gpurun_out/r04/lex_persistent.jsonl | cut -c1-1500
