#!/bin/bash
# round 3, batch 17: several passes in one launch (CCP_GS_MULTI=1) against one launch per pass, with the two-trip landing
OUT=gpurun_out/r03
mkdir -p $OUT
: > $OUT/b17_ab.jsonl
for multi in 0 1 0 1; do
  CCP_GS_MULTI=$multi timeout -k 10 300 python tools/fused_ab.py big mid block region >> $OUT/b17_ab.jsonl 2>> $OUT/b17_ab.err || echo "fused_ab failed for multi=$multi"
done
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r03/b17_ab.jsonl") if l.startswith("{")]
for r in rows:
    if "tuned" in r.get("case","") or "region" in r.get("case","") or r.get("R") in (364,):
        print("multi", r.get("multi"), r.get("case"), "T", r.get("T"), "R", r.get("R"), "ms %.4f" % r.get("ms_per_pass", 0), {k: ("%.3g" % v) for k, v in r.items() if k.startswith("frac")})
PY
