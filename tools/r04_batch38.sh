#!/bin/bash
# round 4 experiment: what the per-sweep step sums cost the checked red-black pass (build without the accumulation: wrong
# stop decisions, same everything else) — kernel stats at 16384^2
cat > /tmp/rb2.py <<'PY'
import os, sys, json; sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from coursecomputationalphotography_amd import capi
g = capi.Grid(16384, 16384, 1); g.randomize_x(1234, 0.0, 255.0); g.b_from_x()
g.fill_x(1.0); g.gauss_seidel(1e-30, 16, 1)
g.fill_x(1.0); rep = g.gauss_seidel(1e-30, 400, 1)[0]
print(json.dumps({"lib": os.environ.get("CCP_GS_LIB", "default"), "iterations": rep.iterations, "seconds": rep.seconds, "updates_per_s": 16384.0 * 16384 * rep.iterations / rep.seconds}))
PY
cd /tmp && export TMPDIR=/tmp
for lib in libccp_gs.so libccp_gs_noacc.so; do
  rm -rf /tmp/prof38
  CCP_GS_LIB=$GRAFT_REPO_ROOT/coursecomputationalphotography_amd/lib/$lib timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof38 -o rb -- python3 /tmp/rb2.py 2>/dev/null | grep "^{"
  f=$(find /tmp/prof38 -name "*kernel_stats.csv" | head -1)
  python3 -c "import csv,sys; [print('  ', r['Name'][:70], r['Calls'], r['AverageNs']) for r in list(csv.DictReader(open(sys.argv[1])))[:4]]" $f
done
