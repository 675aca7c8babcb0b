#!/bin/bash
# round 4: does a launch of 32 groups run slower per sweep than one of 16?  (lex_grid_bench: 256 sweeps fixed 8.6e11, stop rule — batches of 128 — 9.1e11)
mkdir -p gpurun_out/r04
for n in 128 256 512; do
  timeout -k 10 200 python tools/lex_trace.py run 16384 16384 $n gpurun_out/r04/trace_big_$n.bin || exit 1
  python tools/lex_trace.py show gpurun_out/r04/trace_big_$n.bin | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('groups','S','workgroups','launch_us','ns_per_step_running','share_of_slot_time_waiting','mean_workgroups_alive','mean_workgroups_past_their_first_gate','alive_over_time','running_over_time','updates_per_s_of_this_launch')})"
done
