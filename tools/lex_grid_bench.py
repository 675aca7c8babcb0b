#!/usr/bin/env python3
"""Throughput of the reference-order (lexicographic) sweep on the structured grid
(ccp_grid_gauss_seidel_lexicographic: hyperplane pipeline) by size and iteration count, fixed count
and with the reference stop rule after every sweep."""
import json
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from coursecomputationalphotography_amd import capi  # noqa: E402

out = []
for W, H, C, iters in ((512, 512, 1, 100), (4096, 4096, 3, 200), (16384, 16384, 1, 32), (16384, 16384, 1, 256)):
    g = capi.Grid(W, H, C)
    g.randomize_x(1234, 0.0, 255.0)
    g.b_from_x()
    row = {"W": W, "H": H, "channels": C, "iterations": iters}
    for every in (0, 1):
        g.fill_x(1.0)
        g.gauss_seidel_lexicographic(0.0 if every == 0 else 1e-300, min(iters, 8), every)      # warm (allocations)
        g.fill_x(1.0)
        rep = g.gauss_seidel_lexicographic(0.0 if every == 0 else 1e-300, iters, every)[0]
        row["fixed_count" if every == 0 else "stop_rule_every_sweep"] = {
            "seconds": rep.seconds, "updates_per_s": W * H * C * rep.iterations / rep.seconds}
    rr, bb = g.residual_norm2()
    row["rel_residual"] = float((rr[0] / bb[0]) ** 0.5)
    out.append(row)
    print(json.dumps(row), flush=True)
    g.close()
