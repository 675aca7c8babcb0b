#!/bin/bash
# round 3, batch 30: the CPU reference on the 16384^2 system itself (3 sweeps) next to the GPU line
OUT=gpurun_out/r03
mkdir -p $OUT
t0=$(date +%s)
python3 bench.py --cpu-sample 16384 --cpu-iters 3 --no-configs --no-converge --no-reference-order > $OUT/bench_cpu_16384.json 2> $OUT/bench_cpu_16384.err &
pid=$!
while kill -0 $pid 2>/dev/null; do echo "bench.py --cpu-sample 16384 running ($(( $(date +%s) - t0 )) s)"; sleep 30; done
wait $pid; echo "rc=$? after $(( $(date +%s) - t0 )) s"
python3 -c "
import json; r=json.load(open('$OUT/bench_cpu_16384.json')); print(r['value'], r['cpu_baseline'])"
