#!/bin/bash
# round 4: fused conjugate-gradient loop on the stored matrix — parity, then the rate against the three-pass loop
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_csr.py tests/test_gpu_rccl.py tests/test_facade_cpp.py -m gpu -x -q > gpurun_out/r04/cg_tests_b31.log 2>&1
echo "cg tests rc=$?"; tail -3 gpurun_out/r04/cg_tests_b31.log
grep -q " passed" gpurun_out/r04/cg_tests_b31.log || exit 1
grep -q "failed" gpurun_out/r04/cg_tests_b31.log && exit 1
timeout -k 10 600 python - <<'PY' | tee gpurun_out/r04/cg_stored_fused.jsonl
import os, sys, json
sys.path.insert(0, '.')
os.environ["CCP_GS_MASKED"] = "0"
import numpy as np
from coursecomputationalphotography_amd import capi, synth
mask = synth.disc_mask(4096, 4096, seed=4321)
v, c, r, colour, ys, xs = synth.masked_laplacian_csr(mask)
n = len(ys)
m = capi.CsrMatrix().upload_compressed(v, c, r)
b = m.apply_to_vector(synth.x_true(n, 4321))
for mode in ("0", None, "0", None):
    if mode is None: os.environ.pop("CCP_GS_CG_FUSED", None)
    else: os.environ["CCP_GS_CG_FUSED"] = mode
    m.conjugate_gradient(b, 1e-30, 8)
    x, rep = m.conjugate_gradient(b, 1e-30, 200)
    print(json.dumps({"rows": n, "nnz": len(v), "loop": "three-pass" if mode == "0" else "fused", "iterations": rep.iterations,
                      "seconds": rep.seconds, "row_iterations_per_s": n * rep.iterations / rep.seconds,
                      "checksum": float(np.sum(x))}), flush=True)
PY
