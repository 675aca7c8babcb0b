#!/bin/bash
# round 4, call 7: k_lex_wg through several hardware queues sharing one ticket counter — parity, then rates and traces by queue count
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_lex.py tests/test_gpu_region.py -m gpu -x -q > gpurun_out/r04/tests6.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r04/tests6.log
grep -q " passed" gpurun_out/r04/tests6.log || exit 1
rm -f gpurun_out/r04/lex_queues.jsonl
for q in 4 1 2; do
  echo "== CCP_GS_LEX_QUEUES=$q" >> gpurun_out/r04/lex_queues.jsonl
  CCP_GS_LEX_QUEUES=$q timeout -k 10 300 python tools/lex_grid_bench.py >> gpurun_out/r04/lex_queues.jsonl 2>&1
  CCP_GS_LEX_QUEUES=$q timeout -k 10 300 python tools/lex_trace.py run 16384 16384 128 gpurun_out/r04/lex_trace_q$q.bin >> gpurun_out/r04/lex_queues.jsonl 2>&1 && \
  python tools/lex_trace.py show gpurun_out/r04/lex_trace_q$q.bin | tail -1 >> gpurun_out/r04/lex_queues.jsonl 2>&1
done
grep -v amdgpu.ids gpurun_out/r04/lex_queues.jsonl | cut -c1-900
