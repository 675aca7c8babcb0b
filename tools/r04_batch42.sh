#!/bin/bash
# round 4: the mask variant of k_lex_wg at two workgroups per CU — parity, then the region's reference-order rate
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_region.py tests/test_gpu_lex.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r04/region_tests_b42.log 2>&1
echo "tests rc=$?"; tail -2 gpurun_out/r04/region_tests_b42.log
grep -q " passed" gpurun_out/r04/region_tests_b42.log || exit 1
grep -q "failed" gpurun_out/r04/region_tests_b42.log && exit 1
timeout -k 10 600 python - <<'PY'
import sys, json; sys.path.insert(0, '.')
import bench
from coursecomputationalphotography_amd import capi
c = bench.config4(capi)
print(json.dumps({"region": c.get("row_updates_per_s"), "reference_order": c.get("reference_order")}, default=str)[:600])
PY
