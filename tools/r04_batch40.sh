#!/bin/bash
# round 4: small red-black solves, 1,000 iterations without the rule: one pass per launch against several (CCP_GS_MULTI=1)
for multi in 0 1; do
CCP_GS_MULTI=$multi python - <<'PY'
import os, sys, json; sys.path.insert(0, '.')
from coursecomputationalphotography_amd import capi
for W, H, C in ((512, 512, 1), (1024, 1024, 1), (2048, 2048, 1), (4096, 4096, 3)):
    g = capi.Grid(W, H, C); g.randomize_x(1234, 0.0, 255.0); g.b_from_x()
    g.fill_x(1.0); g.gauss_seidel(0.0, 16, 0)
    g.fill_x(1.0)
    rep = g.gauss_seidel(0.0, 1000, 0)[0]
    print(json.dumps({"multi": os.environ["CCP_GS_MULTI"], "W": W, "H": H, "channels": C, "seconds": rep.seconds, "updates_per_s": W * H * C * rep.iterations / rep.seconds}), flush=True)
    g.close()
PY
done
