#!/bin/bash
# round 3, batch 14: prefetch depth of the temporally blocked pass — default build (one trip = 2 rows in flight), two trips
# in flight (land2), the ring form of the window with 4 / 6 rows in flight
OUT=gpurun_out/r03
mkdir -p $OUT
: > $OUT/b14_ab.jsonl
for lib in default land2 ring4 ring6 default land2; do
  if [ $lib = default ]; then unset CCP_GS_LIB; else export CCP_GS_LIB=$PWD/coursecomputationalphotography_amd/lib/libccp_gs_$lib.so; fi
  timeout -k 10 300 python tools/fused_ab.py big mid block >> $OUT/b14_ab.jsonl 2>> $OUT/b14_ab.err || echo "fused_ab failed for $lib"
  echo "done $lib"
done
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r03/b14_ab.jsonl") if l.startswith("{")]
for r in rows:
    print(r.get("lib"), r.get("case"), "T", r.get("T"), "R", r.get("R"), "ms %.4f" % r.get("ms_per_pass", 0), {k: ("%.3g" % v) for k, v in r.items() if k.startswith("frac")})
PY
