#!/bin/bash
# round 4: where a small reference-order solve (512^2, 100 sweeps = configs[0]) spends its 0.9 ms: per-workgroup stamps
mkdir -p gpurun_out/r04
for shape in "512 512 100" "1024 1024 100" "512 512 8"; do
  set -- $shape
  timeout -k 10 120 python tools/lex_trace.py run $1 $2 $3 gpurun_out/r04/trace_$1_$3.bin || exit 1
  python tools/lex_trace.py show gpurun_out/r04/trace_$1_$3.bin | tail -1 | cut -c1-1500
  python tools/lex_trace.py table gpurun_out/r04/trace_$1_$3.bin > gpurun_out/r04/trace_$1_$3.txt
done
head -40 gpurun_out/r04/trace_512_100.txt
