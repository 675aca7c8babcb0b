"""ccp_grid_gauss_seidel_rowblocked with the reference's rule after EVERY sweep, one rank of a (real-RCCL) communicator:
checked temporally blocked passes against the round-2 loop (CCP_GS_ROWBLOCK_CHECKED_FUSED=0: one in-place sweep per check)."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from coursecomputationalphotography_amd import capi
W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
iters = 64
comm = capi.Comm(capi.comm_unique_id(), 0, 1, 0)
for mode in ("1", "0"):
    os.environ["CCP_GS_ROWBLOCK_CHECKED_FUSED"] = mode
    g = capi.Grid(W, H, 1)
    g.randomize_x(1234, 0.0, 255.0)
    g.b_from_x()
    g.attach_comm(comm)
    best = None
    for rep in range(3):
        g.fill_x(1.0)
        r = g.gauss_seidel_rowblocked(1e-300, iters, 1)[0]
        best = r.seconds if best is None else min(best, r.seconds)
    ref = capi.Grid(W, H, 1)
    ref.randomize_x(1234, 0.0, 255.0)
    ref.b_from_x()
    ref.fill_x(1.0)
    r1 = ref.gauss_seidel(1e-300, iters, 1)[0]
    print(json.dumps({"grid": f"{W}x{H}", "iterations": r.iterations, "checked_passes_fused": mode == "1", "seconds": best,
                      "updates_per_s": W * H * iters / best, "one_block_entry_point_updates_per_s": W * H * iters / r1.seconds,
                      "step": r.last_l1_step, "step_one_block": r1.last_l1_step}), flush=True)
    g.attach_comm(None)
    g.close()
    ref.close()
comm.close()
