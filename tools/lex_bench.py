#!/usr/bin/env python3
"""Exact (lexicographic, level-scheduled) Gauss-Seidel on the GPU vs the reference CPU sweep:
BASELINE.json configs[0] (512x512) and larger."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
from coursecomputationalphotography_amd import capi, synth
import oracle
ap = argparse.ArgumentParser(); ap.add_argument("--size", type=int, default=512); ap.add_argument("--iters", type=int, default=100)
a = ap.parse_args()
W = H = a.size
v, c, r = synth.poisson_csr(W, H); b, _ = synth.poisson_system(W, H, 1234)
m = capi.CsrMatrix().upload_compressed(v, c, r)
m.gauss_seidel(b, 0.0, 2, check_every=0, ordering=capi.ORDER_LEXICOGRAPHIC)      # builds the level schedule
x, rep = m.gauss_seidel(b, 0.0, a.iters, check_every=0, ordering=capi.ORDER_LEXICOGRAPHIC)
x1, rep1 = m.gauss_seidel(b, 0.0, a.iters, check_every=1, ordering=capi.ORDER_LEXICOGRAPHIC)
out = {"size": a.size, "iters": a.iters, "gpu_lexicographic_updates_per_s": W * H * a.iters / rep.seconds,
       "gpu_lexicographic_checked_updates_per_s": W * H * a.iters / rep1.seconds, "ms_per_iteration": rep.seconds / a.iters * 1e3}
try:
    ref = oracle.Ref(); secs = ref.gs_csr_timed(v, c, r, b, min(a.iters, 20)); out["cpu_reference_updates_per_s"] = W * H * min(a.iters, 20) / secs
except Exception as e:
    om = oracle.Oracle().from_csr(v, c, r); t0 = time.perf_counter(); want, _, _ = om.gauss_seidel(b, 0.0, min(a.iters, 20)); out["cpu_port_updates_per_s"] = W * H * min(a.iters, 20) / (time.perf_counter() - t0)
print(json.dumps(out))
