#!/bin/bash
# round 4 experiment: where the time between a strip's first gate and its neighbour's first gate goes.  "end" column =
# 1: the strip's first step begins; 2: its second block begins; 3: its 13th block begins (experiment builds, not shipped)
mkdir -p gpurun_out/r04
for k in 1 2 3; do
  echo "== stamp $k"
  CCP_GS_LIB=$PWD/coursecomputationalphotography_amd/lib/libccp_gs_st$k.so timeout -k 10 120 python tools/lex_trace.py run 512 512 8 gpurun_out/r04/trace_x.bin || exit 1
  python tools/lex_trace.py table gpurun_out/r04/trace_x.bin
done
