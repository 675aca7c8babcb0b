#!/bin/bash
# round 4: which kernels the stored-matrix conjugate gradient runs, fused and three-pass (kernel stats)
mkdir -p gpurun_out/r04
cat > /tmp/cgrun.py <<'PY'
import os, sys, json
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
os.environ["CCP_GS_MASKED"] = "0"
import numpy as np
from coursecomputationalphotography_amd import capi, synth
mask = synth.disc_mask(4096, 4096, seed=4321)
v, c, r, colour, ys, xs = synth.masked_laplacian_csr(mask)
n = len(ys)
m = capi.CsrMatrix().upload_compressed(v, c, r)
b = m.apply_to_vector(synth.x_true(n, 4321))
x, rep = m.conjugate_gradient(b, 1e-30, 100)
print(rep.seconds)
PY
cd /tmp && export TMPDIR=/tmp
for mode in 0 1; do
  rm -rf /tmp/prof32
  CCP_GS_CG_FUSED=$mode timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof32 -o cg -- python3 /tmp/cgrun.py > /tmp/p32.log 2>&1
  f=$(find /tmp/prof32 -name "*kernel_stats.csv" | head -1)
  echo "== CCP_GS_CG_FUSED=$mode: $(tail -1 /tmp/p32.log)"
  python3 -c "import csv,sys; [print(r['Name'][:70], r['Calls'], r['AverageNs']) for r in list(csv.DictReader(open(sys.argv[1])))[:8]]" $f
done
