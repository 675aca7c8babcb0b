"""When does the edge flag fire?  One interior row block of an N-GPU run on ONE card (W x H image, H/N owned
rows, `ghost` ghost rows per side): an interval of ghost/2 iterations is issued with
ccp_grid_sweep_edges_first; a side stream waits for the edge flag (ccp_grid_stream_wait_edges) and records
an event.  Reported: interval time with and without the in-launch hand-off, and how far into the LAST pass
the flag fired (0 = at its start, 1 = at its end): the fraction of that pass left for the halo messages.
usage: overlap_probe.py [world=8] [ghost=64] [rank=world//2]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from coursecomputationalphotography_amd import capi  # noqa: E402


def main():
    W = H = 16384
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    ghost = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    rank = int(sys.argv[3]) if len(sys.argv) > 3 else world // 2
    rows = H // world
    g = capi.Grid(W, H, 1, rows * rank, rows, ghost, 0)
    g.randomize_x(1)
    g.b_from_x()
    g.fill_x(1.0)
    k = ghost // 2
    tuned = g.tune(min(8, k // 2))
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()
    res = {}
    for mode in ("plain", "edges"):
        times, fire = [], []
        for rep in range(8):
            g.halo_refreshed()
            e0, e1, ef = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record(main)
            if mode == "edges":
                g.sweep_edges_first(k, ghost)
                g.stream_wait_edges(side.cuda_stream)
                ef.record(side)
            else:
                g.sweep(k)
            e1.record(main)
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
            if mode == "edges":
                fire.append(e0.elapsed_time(ef))
        res[mode] = {"ms_per_interval": min(times[2:])}
        if fire:
            i = min(range(2, 8), key=lambda j: times[j])
            res[mode]["flag_ms_after_interval_start"] = fire[i]
            res[mode]["interval_ms_same_rep"] = times[i]
    n, wait_mode, _, _ = g.comm_stats()
    out = {"world": world, "rank": rank, "block_rows": rows, "ghost": ghost, "iterations_per_interval": k,
           "tuned": tuned, "wait_mode": wait_mode, **res}
    e = res["edges"]
    out["time_left_for_the_exchange_ms"] = e["interval_ms_same_rep"] - e["flag_ms_after_interval_start"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
