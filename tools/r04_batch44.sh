#!/bin/bash
# round 4: the checked red-black loop one pass ahead of the host — whole GPU suite, then the default call's rate
mkdir -p gpurun_out/r04
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04/tests_b44.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r04/tests_b44.log
grep -q " passed" gpurun_out/r04/tests_b44.log || exit 1
grep -q "failed" gpurun_out/r04/tests_b44.log && exit 1
bash tools/r04_batch37.sh 2>&1 | grep "^{" | tee gpurun_out/r04/redblack_default_call.jsonl
