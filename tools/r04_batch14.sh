#!/bin/bash
# round 4: the new full-size band test and the bench's oracle bands, then the whole GPU suite once more on the final tree
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_bench.py -m gpu -x -q > gpurun_out/r04/tests14a.log 2>&1
echo "fullsize+bench tests rc=$?"; tail -3 gpurun_out/r04/tests14a.log
grep -q " passed" gpurun_out/r04/tests14a.log || exit 1
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py --deselect tests/test_gpu_bench.py > gpurun_out/r04/tests14b.log 2>&1
echo "rest rc=$?"; tail -3 gpurun_out/r04/tests14b.log
rm -f gpurun_out/cpu_baseline_phases.log
( time timeout -k 10 900 python bench.py ) > gpurun_out/r04/bench_default5.json 2> gpurun_out/r04/bench_default5.err
echo "bench rc=$?"; tail -3 gpurun_out/r04/bench_default5.err
