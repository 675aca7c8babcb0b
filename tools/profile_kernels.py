#!/usr/bin/env python3
"""Launch the kernels behind BASELINE configs[0] and [4] so that rocprofv3 can see them (kernel trace or
one PMC counter per run; tools/profile_round.sh drives the passes):
  k_sell_apply     SpMV of the 8192^2 mask matrix (applyToVector)
  k_sell_gs        multi-colour Gauss-Seidel sweeps of the same matrix
  k_sell_gs_pipe   the same matrix in the reference's own order (pipelined level schedule)
  k_lex_plane      reference-order sweeps of the 16384^2 grid (hyperplane pipeline)
usage: profile_kernels.py [csr] [lex]   (default: both)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from coursecomputationalphotography_amd import capi, synth  # noqa: E402

what = set(sys.argv[1:]) or {"csr", "lex"}
if "csr" in what:
    mask = synth.disc_mask(8192, 8192, seed=4321)
    v, c, r, colour, ys, xs = synth.masked_laplacian_csr(mask)
    n = len(ys)
    m = capi.CsrMatrix().upload_compressed(v, c, r)
    m.set_colouring(colour, 2)
    xt = synth.x_true(n, 4321)
    b = m.apply_to_vector(xt)
    for _ in range(3):
        m.apply_to_vector(xt)
    m.gauss_seidel(b, 0.0, 10, check_every=0)
    m.gauss_seidel(b, 0.0, 4, check_every=0, ordering=capi.ORDER_LEXICOGRAPHIC)
    m.close()
    print(f"csr: {n} unknowns, {len(v)} non-zeros", flush=True)
if "lex" in what:
    W = H = 16384
    g = capi.Grid(W, H, 1)
    g.randomize_x(1234, 0.0, 255.0)
    g.b_from_x()
    g.fill_x(1.0)
    rep = g.gauss_seidel_lexicographic(0.0, 16, 0)[0]
    print(f"lex: {rep.iterations} sweeps of {W}x{H} in {rep.seconds:.3f} s", flush=True)
    g.close()
