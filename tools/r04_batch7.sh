#!/bin/bash
# round 4: hypothesis test — persistent k_lex_wg WITHOUT the 80-VGPR cap (no vector spills): is the step back at its old speed?
mkdir -p gpurun_out/r04
timeout -k 10 300 python tools/lex_trace.py run 16384 16384 128 gpurun_out/r04/lex_trace_h.bin > gpurun_out/r04/lex_hyp.jsonl 2>&1 && \
python tools/lex_trace.py show gpurun_out/r04/lex_trace_h.bin >> gpurun_out/r04/lex_hyp.jsonl 2>&1
grep -v amdgpu.ids gpurun_out/r04/lex_hyp.jsonl | cut -c1-700
