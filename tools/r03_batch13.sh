#!/bin/bash
# round 3, batch 13: the ring form of the register window (default build) against the shifted form (libccp_gs_shift.so)
OUT=gpurun_out/r03
mkdir -p $OUT
echo "== parity (ring form is the default build)"
timeout -k 10 600 python -m pytest tests/test_gpu_grid.py tests/test_gpu_region.py tests/test_gpu_multi.py -x -q -m gpu > $OUT/b13_tests.log 2>&1; rc=$?; tail -3 $OUT/b13_tests.log
[ $rc -ne 0 ] && { grep -n "^E \|Error" $OUT/b13_tests.log | head -20; exit 1; }
: > $OUT/b13_ab.jsonl
for lib in default shift default shift; do
  if [ $lib = default ]; then unset CCP_GS_LIB; else export CCP_GS_LIB=$PWD/coursecomputationalphotography_amd/lib/libccp_gs_$lib.so; fi
  timeout -k 10 300 python tools/fused_ab.py big mid block region >> $OUT/b13_ab.jsonl 2>> $OUT/b13_ab.err || echo "fused_ab failed for $lib"
  echo "done $lib"
done
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r03/b13_ab.jsonl") if l.startswith("{")]
for r in rows:
    print(r.get("lib"), r.get("case"), "T", r.get("T"), "R", r.get("R"), "ms %.4f" % r.get("ms_per_pass", 0), {k: ("%.3g" % v) for k, v in r.items() if k.startswith("frac") or k.endswith("per_s")})
PY
