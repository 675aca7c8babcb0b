#!/bin/bash
# round 4, call 3: per-workgroup trace of the reference-order sweep at 16384^2 (raw stamps kept)
mkdir -p gpurun_out/r04
timeout -k 10 300 python tools/lex_trace.py run 16384 16384 128 gpurun_out/r04/lex_trace_16384.bin > gpurun_out/r04/lex_trace2.jsonl 2>&1 && \
python tools/lex_trace.py show gpurun_out/r04/lex_trace_16384.bin >> gpurun_out/r04/lex_trace2.jsonl 2>&1
tail -c 3000 gpurun_out/r04/lex_trace2.jsonl
