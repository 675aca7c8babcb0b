"""Whole-image bitwise comparison of the temporally blocked pass with the in-place half-sweep kernels
on unstructured random data (x0 and b), every pixel."""
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def run(W, H, iters, fuse):
    os.environ["CCP_GS_FUSE"] = "1" if fuse else "0"
    from coursecomputationalphotography_amd import capi
    rng = np.random.Generator(np.random.MT19937(5))
    b = rng.uniform(-3.0, 3.0, (H, W))
    x0 = rng.uniform(0.0, 255.0, (H, W))
    g = capi.Grid(W, H, 1)
    g.set_b(b)
    g.set_x(x0)
    g.sweep(iters)
    x = g.get_x().copy()
    g.close()
    return x


if __name__ == "__main__":
    W, H, iters = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (3001, 2003, 16)))
    if len(sys.argv) > 4:
        x = run(W, H, iters, sys.argv[4] == "fused")
        np.save(sys.argv[5], x)
    else:
        out = []
        for mode in ("fused", "inplace"):
            path = f"/tmp/fvi_{mode}.npy"
            subprocess.check_call([sys.executable, __file__, str(W), str(H), str(iters), mode, path])
            out.append(np.load(path))
        same = np.array_equal(out[0], out[1])
        print(f"{W}x{H}, {iters} iterations: fused == in-place on all {W * H} pixels: {same}; "
              f"max |diff| = {np.abs(out[0] - out[1]).max():.3e}")
        sys.exit(0 if same else 1)
