#!/bin/bash
# round 3: the final default bench line (kept as profiles/r03_bench_n1.json), smoke, CPU+GPU suites
OUT=gpurun_out/r03
mkdir -p $OUT
export TMPDIR=/tmp
make -C tests/cpp > /dev/null 2>&1
t0=$(date +%s)
python3 bench.py > $OUT/final_bench_n1.json 2> $OUT/final_bench_n1.err &
pid=$!
while kill -0 $pid 2>/dev/null; do echo "bench.py running ($(( $(date +%s) - t0 )) s)"; sleep 20; done
wait $pid; echo "bench rc=$? after $(( $(date +%s) - t0 )) s"
python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/final_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/final_smoke.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/final_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -4 $OUT/final_gpu_tests.log
