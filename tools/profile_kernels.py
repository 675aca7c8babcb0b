#!/usr/bin/env python3
"""Launch the kernels behind BASELINE configs[0] and [4] so that rocprofv3 can see them (kernel trace or
one PMC counter per run; tools/profile_round.sh drives the passes):
  apply   k_sell_apply     SpMV of the 8192^2 mask matrix (applyToVector)
  gs      k_sell_gs        multi-colour Gauss-Seidel sweeps of the same matrix
  region  k_fused_sweep_masked   the same matrix recognised as a raster region (Dirichlet-mask grid)
  pipe    k_sell_gs_pipe   the same matrix in the reference's own order (pipelined level schedule)
  lex     k_lex_strips     reference-order sweeps of the 16384^2 grid (strip waves; CCP_GS_LEX_MODE=planes: k_lex_plane)
  edge    k_fused_sweep<8,0,2,true>   one interior row block of an 8-GPU run: intervals with the in-launch edge hand-off
  mid     k_fused_sweep<8,0,2>   BASELINE configs[1]: 4096^2 x 3 channels, the tiling bench.py pins (T=8, R=140)
  cg      k_cg_apply_march / k_cg_residual   the fused conjugate-gradient loop on 3 x 8192x4096 (the lab8 blend's call)
usage: profile_kernels.py [--canvas N] [--grid N] [--sweeps K] [apply] [gs] [pipe] [lex]   (default: all four)
Under --pmc the two launch-bound kernels issue tens of thousands of tiny dispatches, each serialised by the
counter collection: use a smaller --canvas / --grid there."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from coursecomputationalphotography_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--canvas", type=int, default=8192)
ap.add_argument("--grid", type=int, default=16384)
ap.add_argument("--sweeps", type=int, default=16)
ap.add_argument("what", nargs="*")
a = ap.parse_args()
what = set(a.what) or {"apply", "gs", "pipe", "lex", "region"}
if what & {"apply", "gs", "pipe", "region"}:
    mask = synth.disc_mask(a.canvas, a.canvas, seed=4321)
    v, c, r, colour, ys, xs = synth.masked_laplacian_csr(mask)
    n = len(ys)
    if "region" in what:                                   # first, while CCP_GS_MASKED is as the caller set it
        m = capi.CsrMatrix().upload_compressed(v, c, r)
        m.set_colouring(colour, 2)
        bb = np.ones(n)
        m.gauss_seidel(bb, 0.0, 16, check_every=0)           # (the second solve tunes the tiling: passes of depth 8)
        m.gauss_seidel(bb, 0.0, 64, check_every=0)
        print(f"region: {m.last_path()}", flush=True)
        m.close()
    os.environ["CCP_GS_MASKED"] = "0"                      # the kernels below are the stored-matrix ones
    m = capi.CsrMatrix().upload_compressed(v, c, r)
    m.set_colouring(colour, 2)
    xt = synth.x_true(n, 4321)
    b = m.apply_to_vector(xt)
    print(f"csr: canvas {a.canvas}, {n} unknowns, {len(v)} non-zeros", flush=True)
    if "apply" in what:
        for _ in range(3):
            m.apply_to_vector(xt)
        print("apply done", flush=True)
    if "gs" in what:
        m.gauss_seidel(b, 0.0, 10, check_every=0)
        print("gs done", flush=True)
    if "pipe" in what:
        m.gauss_seidel(b, 0.0, 4, check_every=0, ordering=capi.ORDER_LEXICOGRAPHIC)
        print("pipe done", flush=True)
    m.close()
if "lex" in what:
    W = H = a.grid
    g = capi.Grid(W, H, 1)
    g.randomize_x(1234, 0.0, 255.0)
    g.b_from_x()
    g.fill_x(1.0)
    rep = g.gauss_seidel_lexicographic(0.0, a.sweeps, 0)[0]
    print(f"lex: {rep.iterations} sweeps of {W}x{H} in {rep.seconds:.3f} s", flush=True)
    g.close()
if "mid" in what:
    g = capi.Grid(4096, 4096, 3)
    g.randomize_x(1234, 0.0, 255.0)
    g.b_from_x()
    g.fill_x(1.0)
    g.set_tiling(8, 140)
    for _ in range(3):
        g.sweep(32)
    g.synchronize()
    print("mid done", flush=True)
    g.close()
if "cg" in what:
    g = capi.Grid(8192, 4096, 3)
    g.randomize_x(1234, 0.0, 255.0)
    g.b_from_x()
    g.fill_x(0.0)
    g.conjugate_gradient(0.0, 10)
    print("cg done", flush=True)
    g.close()
if "edge" in what:
    W = H = 16384
    rows, ghost = H // 8, 64
    g = capi.Grid(W, H, 1, rows * 4, rows, ghost, 0)
    g.randomize_x(1)
    g.b_from_x()
    g.fill_x(1.0)
    g.tune(8)
    for mode in ("plain", "edges"):
        for _ in range(6):
            g.halo_refreshed()
            if mode == "edges":
                g.sweep_edges_first(ghost // 2, ghost)
            else:
                g.sweep(ghost // 2)
            g.synchronize()
    print("edge done", flush=True)
    g.close()
