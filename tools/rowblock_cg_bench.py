"""ccp_grid_conjugate_gradient_rowblocked on one rank of a real-RCCL communicator: the fused loop (72 B, two fetched rows per
neighbour and iteration) against the three-pass loop (88 B), and the one-block entry point."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from coursecomputationalphotography_amd import capi
W, H, C, iters = 8192, 4096, 3, 50
comm = capi.Comm(capi.comm_unique_id(), 0, 1, 0)
for mode in ("1", "0"):
    os.environ["CCP_GS_CG_FUSED"] = mode
    g = capi.Grid(W, H, C)
    g.randomize_x(1234, 0.0, 255.0)
    g.b_from_x()
    g.attach_comm(comm)
    best = best1 = None
    for rep in range(3):
        g.fill_x(0.0)
        r = g.conjugate_gradient_rowblocked(1e-30, iters)
        t = sum(x.seconds for x in r)
        best = t if best is None else min(best, t)
        g.fill_x(0.0)
        g.attach_comm(None)
        r1 = g.conjugate_gradient(1e-30, iters)
        g.attach_comm(comm)
        t1 = sum(x.seconds for x in r1)
        best1 = t1 if best1 is None else min(best1, t1)
    print(json.dumps({"grid": f"{W}x{H}x{C}", "iterations": iters, "loop": "fused (72 B)" if mode == "1" else "three passes (88 B)",
                      "rowblocked_seconds": best, "one_block_seconds": best1, "rowblocked_pixel_iterations_per_s": W * H * C * iters / best}), flush=True)
    g.attach_comm(None)
    g.close()
comm.close()
