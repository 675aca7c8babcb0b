#!/usr/bin/env python3
"""The reference's modify benchmark (labs/lab3/src/OpenCVHW1/main6.cc:92-187) on the device-resident matrix:
a random size x size matrix with `density` % non-zeros, then `modify` % of ALL positions get insert(0)
(removals where an entry exists, no-ops elsewhere), timed.  The reference times its own insert() on the
host; here the same edits go through ccp_csr_insert and the time includes re-laying the touched rows in the
resident images (flushed by an SpMV).  The compiled reference header runs the same scenario beside it
(oracle/_ref, one core) where it has been built.  Also: brush-style edits on the 8192^2 mask matrix —
edit + flush + one sweep, against upload + schedule from scratch."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
from coursecomputationalphotography_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1000)
ap.add_argument("--density", type=int, default=80)
ap.add_argument("--modify", type=int, default=20)
ap.add_argument("--mask-edits", type=int, default=1000)
ap.add_argument("--no-mask", action="store_true")
a = ap.parse_args()
out = {}

# ---- lab3 scenario ---------------------------------------------------------------------------------
rng = np.random.Generator(np.random.MT19937(1))
n = a.size
dense = np.where(rng.integers(0, 100, (n, n)) < a.density, rng.integers(1, 10000, (n, n)), 0).astype(np.float64)
dense[-1, -1] = dense[-1, -1] or 1.0
rows, cols = np.nonzero(dense)
vals = dense[rows, cols]
rowp = np.zeros(n + 1, dtype=np.int32)
np.cumsum(np.bincount(rows, minlength=n), out=rowp[1:])
mods = np.argwhere(rng.integers(0, 100, (n, n)) < a.modify)
m = capi.CsrMatrix()
t0 = time.perf_counter()
m.upload_compressed(vals, cols.astype(np.int32), rowp)
x = np.ones(n)
m.apply_to_vector(x)                                     # builds and uploads the image
t_init = time.perf_counter() - t0
t0 = time.perf_counter()
m.insert_many(np.zeros(len(mods)), mods[:, 0], mods[:, 1])     # one ABI call for the batch (a ctypes call per edit costs ~1 us)
t_edit = time.perf_counter() - t0
t0 = time.perf_counter()
y = m.apply_to_vector(x)                                 # flushes the edits, then multiplies
t_flush = time.perf_counter() - t0
dense[mods[:, 0], mods[:, 1]] = 0.0
ok = bool(np.array_equal(y, dense.sum(axis=1)))          # integer-valued entries: the sums are exact
st = m.edit_stats()
m.close()
out["lab3_modify"] = {"size": n, "density_pct": a.density, "modify_pct": a.modify, "nnz": int(len(vals)), "edits": int(len(mods)),
                      "init_ms": t_init * 1e3, "modify_ms": (t_edit + t_flush) * 1e3, "of_which_host_calls_ms": t_edit * 1e3,
                      "of_which_flush_and_spmv_ms": t_flush * 1e3, "result_equals_dense_mirror": ok, "stats": st}
try:
    import oracle
    ref = oracle.Ref()
    cpu = ref.lab3_modify_bench(rows, cols, vals, mods, n)
    cpu["matches_device"] = bool(cpu["dense_checksum"] == int(dense.sum()))
    out["lab3_modify"]["cpu_reference"] = cpu
except (ImportError, OSError, FileNotFoundError):
    pass

# ---- brush edits on the configs[4] matrix --------------------------------------------------------------
if not a.no_mask:
    mask = synth.disc_mask(8192, 8192, seed=4321)
    v, c, r, colour, ys, xs = synth.masked_laplacian_csr(mask)
    nn = len(ys)
    b = np.ones(nn)
    rows = np.repeat(np.arange(nn, dtype=np.int64), np.diff(r))
    rng = np.random.Generator(np.random.MT19937(5))
    pick = rng.choice(len(v), a.mask_edits, replace=False)
    # (a) the matrix on the general (sliced-ELL) path: every edit is a patch of the resident image.
    # (b) the matrix recognised as a raster region: the FIRST edit ends the matrix-free form (its stencil is no
    #     longer uniform) and the general image is built once; later edits are patches as in (a).
    for label, env in (("general_path", "0"), ("from_region_grid", "1")):
        os.environ["CCP_GS_MASKED"] = env
        m = capi.CsrMatrix()
        t0 = time.perf_counter()
        m.upload_compressed(v, c, r)
        m.set_colouring(colour, 2)
        m.gauss_seidel(b, 0.0, 1, check_every=0)
        t_first = time.perf_counter() - t0
        path0 = m.last_path()
        t0 = time.perf_counter()
        m.insert_many(v[pick] * 0.5, rows[pick], c[pick])
        t_edit = time.perf_counter() - t0
        t0 = time.perf_counter()
        _, rep = m.gauss_seidel(b, 0.0, 1, check_every=0)
        t_solve = time.perf_counter() - t0
        t0 = time.perf_counter()
        m.insert_many(v[pick] * 0.25, rows[pick], c[pick])
        _, rep2 = m.gauss_seidel(b, 0.0, 1, check_every=0)
        t_again = time.perf_counter() - t0
        st = m.edit_stats()
        out["mask_8192_brush_edits_" + label] = {
            "unknowns": nn, "edits_per_batch": int(a.mask_edits), "path_before": path0, "path_after": m.last_path(),
            "upload_setup_first_solve_s": t_first, "edit_calls_ms": t_edit * 1e3,
            "first_batch_flush_plus_one_sweep_incl_host_vectors_ms": t_solve * 1e3,
            "second_batch_edit_flush_sweep_incl_host_vectors_ms": t_again * 1e3, "sweep_device_ms": rep2.seconds * 1e3, "stats": st}
        m.close()
print(json.dumps(out))
