#!/usr/bin/env python3
"""Reference-order (lexicographic) Gauss-Seidel through the GENERAL matrix entry point on the photomontage-style
irregular mask (not SolveChannel's matrix): the raster-region dispatch (the canvas swept in raster order by
k_lex_wg's Dirichlet-mask variant) against the stored-matrix paths (CCP_GS_MASKED=0): the level schedule
pipelined over sweeps (k_sell_gs_pipe) and one launch per level and sweep."""
import argparse, json, os, subprocess, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--canvas", type=int, default=2048)
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--child", default="")
a = ap.parse_args()
if a.child:
    from coursecomputationalphotography_amd import capi, synth
    mask = synth.disc_mask(a.canvas, a.canvas, seed=4321)
    v, c, r, colour, ys, xs = synth.masked_laplacian_csr(mask)
    n = len(ys)
    m = capi.CsrMatrix().upload_compressed(v, c, r)
    m.set_colouring(colour, 2)
    b = m.apply_to_vector(synth.x_true(n, 4321))
    m.gauss_seidel(b, 0.0, 2, check_every=0, ordering=capi.ORDER_LEXICOGRAPHIC)
    x, rep = m.gauss_seidel(b, 0.0, a.iters, check_every=0, ordering=capi.ORDER_LEXICOGRAPHIC)
    x1, rep1 = m.gauss_seidel(b, 1e-300, a.iters, check_every=1, ordering=capi.ORDER_LEXICOGRAPHIC)
    print(json.dumps({"mode": a.child, "canvas": a.canvas, "unknowns": n, "iters": a.iters,
                      "fixed_count_updates_per_s": n * a.iters / rep.seconds,
                      "stop_rule_every_sweep_updates_per_s": n * rep1.iterations / rep1.seconds,
                      "path": m.last_path(), "checksum": float(np.abs(x).sum()), "same_result": bool(np.array_equal(x, x1))}))
else:
    for mode, env in (("region_grid", {}), ("pipelined", {"CCP_GS_MASKED": "0"}),
                      ("one_launch_per_level", {"CCP_GS_MASKED": "0", "CCP_GS_PIPELINE": "0"})):
        iters = a.iters if mode != "one_launch_per_level" else min(a.iters, 5)
        out = subprocess.run([sys.executable, __file__, "--canvas", str(a.canvas), "--iters", str(iters), "--child", mode],
                             env={**os.environ, **env}, capture_output=True, text=True)
        print(out.stdout.strip() or out.stderr[-400:], flush=True)
