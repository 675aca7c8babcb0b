#!/usr/bin/env python3
"""The lab8 panorama-blend workload at full size (coursecomputationalphotography_amd/lab8_workload.py;
reference labs/lab8/src/OpenCVHW1/hw8_pa.cc:749-810): two warped images on a W x H canvas, merged gradient
field, then
  * the reference's own solve — SolveChannel's matrix, 3 channels, start vector = merged colours, 50
    conjugate-gradient iterations (hw8_pa.cc:808-810, 972) — on the structured grid path, and the same
    count of red-black Gauss-Seidel sweeps;
  * the union region as a general CSR matrix (one channel): 50 multi-colour Gauss-Seidel sweeps
    (recognised as a raster region) and, for comparison, the sliced-ELL kernels (CCP_GS_MASKED=0).
Not a bench.py line: a measurement helper."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np  # noqa: E402
from coursecomputationalphotography_amd import capi, lab8_workload as L8  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=8192)
ap.add_argument("--height", type=int, default=4096)
ap.add_argument("--iters", type=int, default=50)
a = ap.parse_args()
W, H, K = a.width, a.height, a.iters
t0 = time.time()
inp = L8.inputs(W, H, 8)
mg = L8.merge(inp)
t_gen = time.time() - t0
out = {"canvas": [W, H], "host_generate_and_merge_s": t_gen}

g = capi.Grid(W, H, 3)
t0 = time.time()
g.assemble_rhs(mg["dx"], mg["dy"], [int(inp["img0"][0, 0, k]) for k in range(3)])
t_asm = time.time() - t0
g.set_x_u8(mg["raw"])
g.tune(8)
g.sweep(2)
g.set_x_u8(mg["raw"])
reps = g.gauss_seidel(0.0, K, 0)
rr, bb = g.residual_norm2()
out["full_canvas_red_black_gs"] = {"iterations": K, "ms": reps[0].seconds * 1e3, "pixel_updates_per_s": W * H * 3 * K / reps[0].seconds,
                                   "rel_residual": [float(np.sqrt(a_ / b_)) for a_, b_ in zip(rr, bb)]}
g.set_x_u8(mg["raw"])
reps = g.conjugate_gradient(1e-10, K)
secs = sum(r.seconds for r in reps)
rr, bb = g.residual_norm2()
res = g.store_u8()
out["full_canvas_cg_reference_call"] = {"iterations": [r.iterations for r in reps], "ms": secs * 1e3,
                                        "pixel_iterations_per_s": W * H * sum(r.iterations for r in reps) / secs,
                                        "rel_residual": [float(np.sqrt(a_ / b_)) for a_, b_ in zip(rr, bb)],
                                        "assemble_rhs_incl_pcie_ms": t_asm * 1e3, "result_mean": float(res.mean())}
g.close()

rv, rc, rr_, colour, ys, xs, b, x0 = L8.region_system(mg, 1)
n = len(ys)
for label, env in (("region_grid", "1"), ("sliced_ell", "0")):
    os.environ["CCP_GS_MASKED"] = env
    m = capi.CsrMatrix().upload_compressed(rv, rc, rr_)
    m.set_colouring(colour, 2)
    t0 = time.time()
    m.gauss_seidel(b, 0.0, 2, x0=x0, check_every=0)
    t_first = time.time() - t0
    x, rep = m.gauss_seidel(b, 0.0, K, x0=x0, check_every=0)
    r2, b2 = m.residual_norm2(b, x)
    out["union_region_" + label] = {"unknowns": n, "path": m.last_path(), "iterations": K, "ms_per_iteration": rep.seconds * 1e3 / K,
                                    "row_updates_per_s": n * K / rep.seconds, "first_solve_incl_setup_s": t_first,
                                    "rel_residual": float(np.sqrt(r2 / b2)), "checksum": float(np.abs(x).sum())}
    # the same region through the reference's unchanged call: gaussSeidel in index order (the facade's default)
    m.gauss_seidel(b, 0.0, 2, x0=x0, check_every=0, ordering=capi.ORDER_LEXICOGRAPHIC)
    x, rep = m.gauss_seidel(b, 0.0, K, x0=x0, check_every=0, ordering=capi.ORDER_LEXICOGRAPHIC)
    out["union_region_" + label]["reference_order"] = {"path": m.last_path(), "iterations": K, "row_updates_per_s": n * K / rep.seconds,
                                                       "checksum": float(np.abs(x).sum())}
    m.close()
out["union_region_paths_agree"] = (out["union_region_region_grid"]["checksum"] == out["union_region_sliced_ell"]["checksum"]
                                   and out["union_region_region_grid"]["reference_order"]["checksum"]
                                   == out["union_region_sliced_ell"]["reference_order"]["checksum"])
print(json.dumps(out))
