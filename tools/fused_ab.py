"""A/B of the temporally blocked pass (k_fused_sweep) on the shapes the bench quotes: average pass time over
a timed region (HIP events on the launch stream) for
  * 16384^2 x 1, pinned tiling (T=8, R=364) and a few chunk heights,
  * 4096^2 x 3 (BASELINE configs[1]) — tuned, and a sweep of chunk heights at T=8,
  * one interior 2048-row block of an 8-GPU run (ghost 64),
  * the 8192^2 region grid (BASELINE configs[4]) when --region is given.
Run it once per build / environment (CCP_GS_LIB, CCP_GS_XCD); one JSON line per case on stdout."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from coursecomputationalphotography_amd import capi  # noqa: E402

TAG = {"lib": os.path.basename(os.environ.get("CCP_GS_LIB", "default")), "multi": os.environ.get("CCP_GS_MULTI", "0")}


def timed(g, iters, steps=6):
    g.sweep(iters)
    g.halo_refreshed()
    g.region_begin()
    for _ in range(steps):
        g.sweep(iters)
        g.halo_refreshed()
    ms, launches, its = g.region_end()
    return ms / max(launches, 1), launches


def case(name, W, H, C, T, R, iters, row_begin=0, rows=None, ghost=0):
    g = capi.Grid(W, H, C, row_begin, rows, ghost, 0)
    g.randomize_x(1234, 0.0, 255.0)
    g.b_from_x()
    g.fill_x(1.0)
    g.halo_refreshed()
    if R == 0:
        T, R, _ = g.tune(T)
    else:
        g.set_tiling(T, R)
    ms, launches = timed(g, iters)
    own = rows if rows else H
    px = float(W) * own * C
    print(json.dumps({**TAG, "case": name, "T": T, "R": R, "ms_per_pass": ms, "launches": launches,
                      "updates_per_s": px * T / (ms * 1e-3), "frac_24B": 24.0 * px / (ms * 1e-3) / 8e12}), flush=True)
    g.close()


def main():
    which = sys.argv[1:] or ["big", "mid", "block"]
    if "big" in which:
        for R in (364, 496, 256, 728):
            case("16384x16384x1", 16384, 16384, 1, 8, R, 32)
    if "mid" in which:
        case("4096x4096x3 tuned", 4096, 4096, 3, 8, 0, 32)
        for R in (96, 140, 184, 256, 274, 342, 404, 512):
            case("4096x4096x3", 4096, 4096, 3, 8, R, 32)
    if "block" in which:
        case("16384x2048 block of 8 (ghost 64) tuned", 16384, 16384, 1, 8, 0, 32, 2048 * 4, 2048, 64)
        for R in (128, 198, 256, 344, 520):
            case("16384x2048 block of 8 (ghost 64)", 16384, 16384, 1, 8, R, 32, 2048 * 4, 2048, 64)
    if "region" in which:
        from coursecomputationalphotography_amd import synth
        mask = synth.disc_mask(8192, 8192, seed=4321)
        for T in (8, 7, 6):
            g = capi.Grid(8192, 8192, 1, mask=mask)
            g.randomize_x(1, 0.0, 255.0)
            g.b_from_x()
            g.fill_x(1.0)
            Tt, R, _ = g.tune(T)
            ms, launches = timed(g, 4 * Tt)
            n = int(mask.sum())
            print(json.dumps({**TAG, "case": "8192^2 disc mask region grid", "T": Tt, "R": R, "ms_per_pass": ms,
                              "row_updates_per_s": n * Tt / (ms * 1e-3), "frac_25B": 25.0 * n / (ms * 1e-3) / 8e12}), flush=True)
            g.close()


if __name__ == "__main__":
    main()
