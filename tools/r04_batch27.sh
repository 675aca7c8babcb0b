#!/bin/bash
# round 4: host-side laps of a 256-sweep reference-order call (CCP_GS_DEBUG)
mkdir -p gpurun_out/r04
CCP_GS_DEBUG=1 timeout -k 10 200 python - <<'PY' 2>&1 | grep -v "^\[ccp_gs\] k_lex_wg" | tail -40
import sys; sys.path.insert(0,'.')
from coursecomputationalphotography_amd import capi
g=capi.Grid(16384,16384,1); g.randomize_x(1234,0.0,255.0); g.b_from_x()
for n in (8,32,256,256):
    g.fill_x(1.0)
    print("== sweeps",n,flush=True)
    rep=g.gauss_seidel_lexicographic(0.0,n,0)[0]
    print("   seconds",rep.seconds, 16384*16384*n/rep.seconds,flush=True)
PY
