#!/usr/bin/env python3
"""conjugateGradient on the structured grid: pixel-iterations/s and algorithmic
GB/s on one MI355X.  Algorithmic bytes per pixel per CG iteration (fp64, matrix-free): three-pass loop
(CCP_GS_CG_FUSED=0) SpMV read p 8 + write Ap 8 (p'Ap fused in); x,r update read 32 + write 16; direction read 16 +
write 8 => 88 B; fused loop (default, csrc/ccp_grid_cg.hpp) x rw 16 + r 8 + p r/w 16 + Ap w 8, then r rw 16 + Ap 8 => 72 B."""
import argparse, json, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
from coursecomputationalphotography_amd import capi

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=8192)
ap.add_argument("--channels", type=int, default=1)
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--height", type=int, default=0)
a = ap.parse_args()
H = a.height or a.size
fused = os.environ.get("CCP_GS_CG_FUSED", "1") != "0"
g = capi.Grid(a.size, H, a.channels)
g.randomize_x(1234, 0.0, 255.0)
g.b_from_x()
g.fill_x(0.0)
g.conjugate_gradient(0.0, 5)                       # warm
g.fill_x(0.0)
reps = g.conjugate_gradient(0.0, a.iters)
secs = sum(r.seconds for r in reps)
rr, bb = g.residual_norm2()
n = a.size * H * a.channels
model = 72.0 if fused else 88.0
print(json.dumps({"width": a.size, "height": H, "channels": a.channels, "iters": a.iters, "loop": "fused (72 B)" if fused else "three passes (88 B)",
                  "seconds": secs, "pixel_iterations_per_s": n * a.iters / secs,
                  "algorithmic_GBps": model * n * a.iters / secs / 1e9, "frac_of_8TBps": model * n * a.iters / secs / 8e12,
                  "rel_residual": float(np.sqrt(rr / bb).max())}))
