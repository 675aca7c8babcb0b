#!/bin/bash
# round 4: the multi-colour (red-black) solve with the reference's defaults (epsilon 1e-6, 1,000 iterations, rule after
# every iteration) against the same count without the rule, and its kernels (stats)
mkdir -p gpurun_out/r04
cat > /tmp/rb.py <<'PY'
import os, sys, json; sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from coursecomputationalphotography_amd import capi
for W, H, C in ((512, 512, 1), (4096, 4096, 3), (16384, 16384, 1)):
    g = capi.Grid(W, H, C); g.randomize_x(1234, 0.0, 255.0); g.b_from_x()
    g.fill_x(1.0); g.gauss_seidel(1e-6, 16, 1); g.fill_x(1.0); g.gauss_seidel(0.0, 16, 0)
    for every, eps, n in ((1, 1e-6, 1000), (8, 1e-6, 1000), (0, 0.0, 1000)):
        g.fill_x(1.0)
        rep = g.gauss_seidel(eps, n, every)[0]
        print(json.dumps({"W": W, "H": H, "channels": C, "check_every": every, "iterations": rep.iterations, "seconds": rep.seconds,
                          "updates_per_s": W * H * C * rep.iterations / rep.seconds}), flush=True)
    g.close()
PY
python3 /tmp/rb.py | tee gpurun_out/r04/redblack_default_call.jsonl
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof37
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof37 -o rb -- python3 /tmp/rb.py > /tmp/p37.log 2>&1
f=$(find /tmp/prof37 -name "*kernel_stats.csv" | head -1)
python3 -c "import csv,sys; [print(r['Name'][:90], r['Calls'], r['AverageNs'], r['Percentage']) for r in list(csv.DictReader(open(sys.argv[1])))[:12]]" $f
