#!/bin/bash
# round 4: per-strip run time against the image height (one group): slope = the inner step, intercept = head and tail
mkdir -p gpurun_out/r04
for H in 512 1024 2048; do
  timeout -k 10 120 python tools/lex_trace.py run 512 $H 8 gpurun_out/r04/trace_h$H.bin || exit 1
  python tools/lex_trace.py table gpurun_out/r04/trace_h$H.bin > gpurun_out/r04/trace_h$H.txt
  cat gpurun_out/r04/trace_h$H.txt
done
