#!/bin/bash
# round 4: sliced-ELL kernels with batched entry loads: the CG loops again, and configs[4]'s stored-matrix leg
mkdir -p gpurun_out/r04
sed -n '/^timeout -k 10 600 python - <<.PY. | tee/,/^PY$/p' tools/r04_batch31.sh > /tmp/cgpart.sh; bash /tmp/cgpart.sh
timeout -k 10 600 python - <<'PY'
import sys, json; sys.path.insert(0, '.')
import bench
from coursecomputationalphotography_amd import capi
c = bench.config4(capi)
print(json.dumps({"region": c.get("row_updates_per_s"), "general_csr_path": c.get("general_csr_path")}, default=str)[:900])
PY
