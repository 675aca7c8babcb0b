"""What the drop-in call pays for handing host vectors over (VERDICT r2 "the facade hand-off is
PCIe-bound"): wall time of ccp_grid_set_b_host / ccp_grid_get_x_host at 16384^2 (2.15 GB each way) and of a whole
ccp_csr_gauss_seidel call (b in, 50 sweeps, x out) at the 8192^2 region matrix, against the device time of the sweeps.
One JSON line per measurement."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from coursecomputationalphotography_amd import capi, synth  # noqa: E402


def say(**kw):
    print(json.dumps(kw), flush=True)


def grid(n):
    g = capi.Grid(n, n, 1)
    host = np.random.default_rng(1).uniform(0, 255, (n, n))
    out = None
    for rep in range(3):
        t = time.perf_counter()
        g.set_b(host)
        g.synchronize()
        t_in = time.perf_counter() - t
        t = time.perf_counter()
        out = g.get_b()
        t_out = time.perf_counter() - t
        say(what=f"{n}x{n} grid", rep=rep, bytes=host.nbytes, set_b_host_s=t_in, to_device_GBps=host.nbytes / t_in / 1e9,
            get_host_s=t_out, to_host_GBps=host.nbytes / t_out / 1e9, round_trip_equal=bool(np.array_equal(out, host)))
    # the same copy into memory that has been touched before (a std::vector<double> x(n) of the C++ caller): no page faults
    import ctypes as C
    dst = np.zeros((n, n))
    for rep in range(3):
        t = time.perf_counter()
        capi.check(g.L.ccp_grid_get_b_host(g.h, 0, dst.ctypes.data_as(C.c_void_p), 0, n), "ccp_grid_get_b_host")
        t_out = time.perf_counter() - t
        say(what=f"{n}x{n} grid, destination touched before", rep=rep, get_host_s=t_out, to_host_GBps=host.nbytes / t_out / 1e9,
            equal=bool(np.array_equal(dst, host)))
    g.fill_x(1.0)
    g.sweep(32)
    g.synchronize()
    t = time.perf_counter()
    g.sweep(400)
    g.synchronize()
    say(what=f"{n}x{n} grid", sweeps_400_s=time.perf_counter() - t)
    g.close()


def region():
    mask = synth.disc_mask(8192, 8192)
    v, col, rowp, colour, _, _ = synth.masked_laplacian_csr(mask)
    n = len(rowp) - 1
    b = synth.csr_apply(v, col, rowp, synth.x_true(n, 1234))
    m = capi.CsrMatrix().upload_compressed(v, col, rowp).set_colouring(colour, 2)
    for rep in range(4):
        t = time.perf_counter()
        x, r = m.gauss_seidel(b, 0.0, 50, check_every=0)
        wall = time.perf_counter() - t
        say(what="8192^2 region matrix, ccp_csr_gauss_seidel(50 sweeps)", rep=rep, unknowns=n, wall_s=wall, sweeps_device_s=r.seconds,
            hand_over_s=wall - r.seconds, vector_bytes_each_way=8 * n, path=m.last_path())
    out = np.zeros(n)
    for rep in range(4):
        t = time.perf_counter()
        x, r = m.gauss_seidel(b, 0.0, 50, check_every=0, out=out)
        wall = time.perf_counter() - t
        say(what="the same call, x into a buffer that is used again", rep=rep, wall_s=wall, sweeps_device_s=r.seconds, hand_over_s=wall - r.seconds)
    m.close()


if __name__ == "__main__":
    what = sys.argv[1:] or ["grid", "region"]
    if "grid" in what:
        grid(16384)
    if "region" in what:
        region()
