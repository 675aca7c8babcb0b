#!/usr/bin/env python3
"""conjugateGradient on the structured grid: pixel-iterations/s and algorithmic
GB/s on one MI355X.  Algorithmic bytes per pixel per CG iteration (fp64, matrix-free):
SpMV read p 8 + write Ap 8; p'Ap read 16 (0 when fused into the SpMV); x,r update read 32 + write 16;
direction read 16 + write 8  => 104 B unfused, 88 B with the dot fused."""
import argparse, json, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
from coursecomputationalphotography_amd import capi

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=8192)
ap.add_argument("--channels", type=int, default=1)
ap.add_argument("--iters", type=int, default=50)
a = ap.parse_args()
g = capi.Grid(a.size, a.size, a.channels)
g.randomize_x(1234, 0.0, 255.0)
g.b_from_x()
g.fill_x(0.0)
g.conjugate_gradient(0.0, 5)                       # warm
g.fill_x(0.0)
reps = g.conjugate_gradient(0.0, a.iters)
secs = sum(r.seconds for r in reps)
rr, bb = g.residual_norm2()
n = a.size * a.size * a.channels
print(json.dumps({"size": a.size, "channels": a.channels, "iters": a.iters, "seconds": secs,
                  "pixel_iterations_per_s": n * a.iters / secs,
                  "algorithmic_GBps_at_104B": 104.0 * n * a.iters / secs / 1e9,
                  "rel_residual": float(np.sqrt(rr / bb).max())}))
