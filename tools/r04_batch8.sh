#!/bin/bash
# round 4, call: full GPU suite on the persistent reference-order kernel; kernel trace of the reference-order bench (what the
# conversions to and from the diagonal-major layout cost beside the sweep)
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04/tests8.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r04/tests8.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04/kt_lex -o k -- python3 $GRAFT_REPO_ROOT/tools/lex_grid_bench.py > $GRAFT_REPO_ROOT/gpurun_out/r04/kt_lex.txt 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r04/kt_lex -name "*kernel_stats*" | head -2
f=$(find gpurun_out/r04/kt_lex -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-220
