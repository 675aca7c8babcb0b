#!/usr/bin/env python3
"""Throughput of ccp_grid_gauss_seidel at 16384^2 by check_every (reference stop rule evaluated
every k-th sweep): 0 = never, 1 = every sweep (reference behaviour), 8 = every 8th."""
import json, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from coursecomputationalphotography_amd import capi
W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
g = capi.Grid(W, H, 1)
g.randomize_x(1234, 0.0, 255.0); g.b_from_x(); g.fill_x(1.0)
g.tune(8)
out = {}
for every in (0, 8, 1):
    g.fill_x(1.0)
    g.gauss_seidel(0.0 if every == 0 else 1e-300, 40, every)          # warm
    g.fill_x(1.0)
    rep = g.gauss_seidel(0.0 if every == 0 else 1e-300, 120, every)[0]
    out[f"check_every_{every}"] = {"iterations": rep.iterations, "updates_per_s": W * H * rep.iterations / rep.seconds,
                                   "last_l1_step": rep.last_l1_step}
os.environ["CCP_GS_FUSE"] = "0"
g2 = capi.Grid(W, H, 1); g2.randomize_x(1234, 0.0, 255.0); g2.b_from_x(); g2.fill_x(1.0)
rep = g2.gauss_seidel(1e-300, 40, 1)[0]
out["check_every_1_in_place_kernels"] = {"iterations": rep.iterations, "updates_per_s": W * H * rep.iterations / rep.seconds,
                                         "last_l1_step": rep.last_l1_step}
print(json.dumps(out))
