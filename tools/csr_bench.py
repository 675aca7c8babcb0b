#!/usr/bin/env python3
"""BASELINE.json configs[4]: 8192x8192 canvas, photomontage-style irregular mask, general CSR
path on one MI355X.  Reports rows/s and algorithmic GB/s (12*nnz + 32 B per row update,
SURVEY §8d) for the multi-colour Gauss-Seidel sweep and the SpMV, plus a fixed-point parity
property (x_true stays a fixed point).  Not a bench.py line: a measurement helper."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
from coursecomputationalphotography_amd import capi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--canvas", type=int, default=8192)
ap.add_argument("--iters", type=int, default=50)
a = ap.parse_args()
t0 = time.time()
mask = synth.disc_mask(a.canvas, a.canvas, seed=4321)
v, c, r, colour, ys, xs = synth.masked_laplacian_csr(mask)
n, nnz = len(ys), len(v)
t_gen = time.time() - t0
m = capi.CsrMatrix()
t0 = time.time(); m.upload_compressed(v, c, r); m.set_colouring(colour, 2); t_up = time.time() - t0
xt = synth.x_true(n, 4321)
t0 = time.time(); b = m.apply_to_vector(xt); t_spmv_first = time.time() - t0     # builds the natural schedule
t0 = time.time(); x, rep = m.gauss_seidel(b, 0.0, 2, check_every=0); t_first = time.time() - t0   # builds the colour schedule
x, rep = m.gauss_seidel(b, 0.0, a.iters, check_every=0)
bytes_per_iter = 12.0 * nnz + 32.0 * n
x2, rep2 = m.gauss_seidel(b, 0.0, 4, x0=xt, check_every=0)
rr, bb = m.residual_norm2(b, x2)
out = {"canvas": a.canvas, "unknowns": n, "nnz": nnz, "iters": a.iters,
       "gs_seconds": rep.seconds, "row_updates_per_s": n * a.iters / rep.seconds,
       "algorithmic_GBps": bytes_per_iter * a.iters / rep.seconds / 1e9,
       "frac_of_8TBps": bytes_per_iter * a.iters / rep.seconds / 8e12,
       "fixed_point_rel_residual": float(np.sqrt(rr / bb)),
       "host_seconds": {"generate": t_gen, "upload": t_up, "first_spmv_incl_schedule": t_spmv_first,
                        "first_solve_incl_schedule": t_first}}
print(json.dumps(out))
