#!/bin/bash
# round 4: the stop-rule path with growing batches — parity, then the reference's default call (epsilon 1e-6, 1,000 sweeps)
mkdir -p gpurun_out/r04
timeout -k 10 300 python -m pytest tests/test_gpu_lex.py tests/test_gpu_region.py tests/test_facade_cpp.py -m gpu -x -q > gpurun_out/r04/lex_tests_b35.log 2>&1
echo "lex tests rc=$?"; tail -2 gpurun_out/r04/lex_tests_b35.log
grep -q " passed" gpurun_out/r04/lex_tests_b35.log || exit 1
grep -q "failed" gpurun_out/r04/lex_tests_b35.log && exit 1
timeout -k 10 300 python - <<'PY' | tee gpurun_out/r04/lex_default_call.jsonl
import sys, json; sys.path.insert(0, '.')
from coursecomputationalphotography_amd import capi
for W, H, C in ((512, 512, 1), (4096, 4096, 3), (16384, 16384, 1)):
    g = capi.Grid(W, H, C); g.randomize_x(1234, 0.0, 255.0); g.b_from_x()
    g.fill_x(1.0); g.gauss_seidel_lexicographic(1e-6, 8, 1)
    for every, eps in ((1, 1e-6), (0, 0.0)):
        g.fill_x(1.0)
        rep = g.gauss_seidel_lexicographic(eps, 1000, every)[0]
        print(json.dumps({"W": W, "H": H, "channels": C, "call": "gaussSeidel(b): epsilon 1e-6, 1000 sweeps, rule after every sweep" if every else "1000 sweeps, fixed count",
                          "iterations": rep.iterations, "converged": rep.converged, "seconds": rep.seconds,
                          "updates_per_s": W * H * C * rep.iterations / rep.seconds}), flush=True)
    g.close()
PY
