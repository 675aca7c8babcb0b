"""First colour-ordered solve of the 8192^2 region matrix WITHOUT a colouring from the caller (what the facade does):
upload, first solve (colouring + recognition + 2 sweeps), second solve."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from coursecomputationalphotography_amd import capi, synth
mask = synth.disc_mask(8192, 8192)
v, c, r, colour, _, _ = synth.masked_laplacian_csr(mask)
n = len(r) - 1
b = np.ones(n)
order = capi.ORDER_LEXICOGRAPHIC if len(sys.argv) > 1 and sys.argv[1] == "lex" else capi.ORDER_MULTICOLOUR
for given in (True, False):
    m = capi.CsrMatrix()
    t = time.perf_counter(); m.upload_compressed(v, c, r); t_up = time.perf_counter() - t
    if given:
        m.set_colouring(colour, 2)
    t = time.perf_counter(); m.gauss_seidel(b, 0.0, 2, check_every=0, ordering=order); t1 = time.perf_counter() - t
    t = time.perf_counter(); m.gauss_seidel(b, 0.0, 2, check_every=0, ordering=order); t2 = time.perf_counter() - t
    print(json.dumps({"ordering": "reference" if order == capi.ORDER_LEXICOGRAPHIC else "colour", "colouring_given": given, "upload_s": round(t_up, 3), "first_solve_s": round(t1, 3), "second_solve_s": round(t2, 3), "path": m.last_path()}), flush=True)
    m.close()
