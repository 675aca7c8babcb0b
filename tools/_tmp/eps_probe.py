import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import oracle
from coursecomputationalphotography_amd import capi, synth
W, H = 97, 61
b, _ = synth.poisson_system(W, H, 5); b = b * 1e-3
m = oracle.Oracle().from_csr(*synth.poisson_csr(W, H))
for k in list(range(1, 18)) + [63, 64, 120, 126, 127, 128]:
    _, it, e = m.gauss_seidel(b, 0.0, k)
    g = capi.Grid(W, H, 1); g.set_b(b); g.fill_x(1.0)
    rep = g.gauss_seidel_lexicographic(1e-300, k, 1)[0]
    x = g.get_x().ravel(); g.close()
    want = m.gauss_seidel(b, 0.0, k)[0]
    print(k, "eps dev %.17g ora %.17g rel %.3e  x_equal %s" % (rep.last_l1_step, e, (rep.last_l1_step - e) / e, np.array_equal(x, want)), flush=True)
