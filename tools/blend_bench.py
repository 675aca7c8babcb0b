#!/usr/bin/env python3
"""BASELINE configs[1] end to end through the image-side entry points:
4096x4096 3-channel blend of 4 source images under a label map —
  ccp_grid_assemble_from_images (GradientAt + ATb + composite start vector, PhotoMontage.cpp:399-436,541-610)
  -> Gauss-Seidel, fixed count (red-black; reference order; conjugate gradient)
  -> ccp_grid_store_u8 (clamp epilogue, PhotoMontage.cpp:617-626).
Host <-> device image copies are inside the assemble / store figures (they take host pointers)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from coursecomputationalphotography_amd import capi  # noqa: E402

W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.Generator(np.random.MT19937(3))
# smooth-ish sources (low-frequency ramps + noise), label map of 4 vertical/horizontal regions
yy, xx = np.mgrid[0:H, 0:W]
images = []
for k in range(4):
    base = (np.sin(xx / (97.0 + 13 * k)) * 60 + np.cos(yy / (71.0 + 7 * k)) * 50 + 128 + 10 * k)
    img = np.clip(base[..., None] + rng.normal(0, 4, (H, W, 3)), 0, 255).astype(np.uint8)
    images.append(img)
label = ((xx * 2 // W) + 2 * (yy * 2 // H)).astype(np.uint8)
out = {"W": W, "H": H, "channels": 3, "iterations": iters}
g = capi.Grid(W, H, 3)
t0 = time.perf_counter()
g.assemble_from_images(images, label, init_x=True)
g.synchronize()
out["assemble_from_images_s"] = time.perf_counter() - t0
g.tune(8)
for name in ("red_black", "reference_order", "conjugate_gradient"):
    g.assemble_from_images(images, label, init_x=True)
    g.synchronize()
    if name == "red_black":
        rep = g.gauss_seidel(1e-10, iters, 0)[0]
    elif name == "reference_order":
        rep = g.gauss_seidel_lexicographic(1e-10, iters, 0)[0]
    else:
        rep = g.conjugate_gradient(1e-10, iters)[0]
    rr, bb = g.residual_norm2()
    out[name] = {"solve_s": rep.seconds, "pixel_iterations_per_s": W * H * 3 * rep.iterations / rep.seconds,
                 "rel_residual": [float(v) for v in np.sqrt(rr / bb)]}
t0 = time.perf_counter()
res = g.store_u8()
out["store_u8_s"] = time.perf_counter() - t0
out["result_mean"] = float(res.mean())
print(json.dumps(out))
