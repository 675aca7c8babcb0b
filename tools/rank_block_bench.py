"""Time the sweeps of ONE interior row block of a multi-GPU run on one card: W x H image, the block
of `rows` owned rows starting at `row_begin`, `ghost` ghost rows per side; ghost/2 iterations per
interval, halos declared refreshed (no neighbour here — the values are irrelevant for timing).

  python tools/rank_block_bench.py [world ghost plain|edges rank [mask]]
mask: the block is a row block of the Dirichlet-mask grid of BASELINE configs[4] (8192^2 canvas, discs + brush trail)."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from coursecomputationalphotography_amd import capi  # noqa: E402


def main():
    W = H = 16384
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    ghost = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    edges_first = (sys.argv[3] == "edges") if len(sys.argv) > 3 else False
    rank = int(sys.argv[4]) if len(sys.argv) > 4 else world // 2
    mask = None
    if len(sys.argv) > 5 and sys.argv[5] == "mask":
        from coursecomputationalphotography_amd import synth
        W = H = 8192
        mask = synth.disc_mask(W, H).astype("uint8")
    rows = H // world
    g = capi.Grid(W, H, 1, rows * rank, rows, ghost, 0, mask=mask)
    g.randomize_x(1)
    g.b_from_x()
    g.fill_x(1.0)
    k = ghost // 2
    tuned = g.tune(min(8, k // 2))
    g.halo_refreshed()
    out = []
    for rep in range(6):
        if edges_first:
            g.sweep_edges_first(k, ghost)
        else:
            g.sweep(k)
        g.synchronize()
        ms, launches = g.last_timing()
        out.append(ms)
        g.halo_refreshed()
    best = min(out[1:])
    res = {"world": world, "rank": rank, "block_rows": rows, "ghost": ghost, "iterations_per_interval": k,
           "tuned": tuned, "ms_per_interval": best, "edges_first": edges_first}
    if mask is None:
        res.update({"block_updates_per_s": W * rows * k / best * 1e3, "node_updates_per_s_if_exchange_hidden": W * H * k / best * 1e3})
    else:
        own = int(mask[rows * rank:rows * (rank + 1)].sum())
        res.update({"mask": "8192^2 discs + brush", "unknowns_in_block": own, "unknowns": int(mask.sum()),
                    "block_row_updates_per_s": own * k / best * 1e3,
                    "halo_bytes_per_interval": 2 * ghost * 2 * (((W + 1) // 2 + 15) // 16 * 16) * 8})
    print(json.dumps(res))


if __name__ == "__main__":
    main()
