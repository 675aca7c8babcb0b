"""BASELINE configs[4] (8192^2 irregular region, general CSR path) in ROW BLOCKS (ccp_csr_upload_rows).

  python tools/csr_rows_bench.py block [world]     one block of `world` (default 8) on this GPU through the real RCCL at
                                                   world 1: its couplings to other blocks are cut, so the sweep does the
                                                   block's arithmetic without its messages -> per-GPU kernel rate, plus
                                                   the halo sizes of every block of the partition (numpy)
  python tools/csr_rows_bench.py threads [world]   all `world` blocks as threads on this ONE card over the test transport
                                                   (CCP_GS_RCCL_LIB=tests/cpp/libfake_rccl.so): full-size parity with
                                                   the one-GPU handle, messages per sweep; the time is NOT a multi-GPU time
One JSON line per measurement on stdout.
"""
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from coursecomputationalphotography_amd import capi, synth  # noqa: E402

N = int(os.environ.get("CSR_ROWS_CANVAS", "8192"))
SWEEPS = 50


def say(**kw):
    print(json.dumps(kw), flush=True)


def system():
    t = time.time()
    mask = synth.disc_mask(N, N)
    v, col, rowp, colour, _, _ = synth.masked_laplacian_csr(mask)
    n = len(rowp) - 1
    say(stage="matrix", canvas=N, unknowns=n, entries=int(rowp[-1]), seconds=round(time.time() - t, 1))
    return v, col, rowp.astype(np.int64), colour, n


def cuts_for(n, world):
    return [int(round(n * k / world)) for k in range(world + 1)]


def halo_sizes(col, rowp, cuts):
    out = []
    for k in range(len(cuts) - 1):
        lo, hi = cuts[k], cuts[k + 1]
        c = col[rowp[lo]:rowp[hi]]
        g = np.unique(c[(c < lo) | (c >= hi)])
        out.append(int(len(g)))
    return out


def mode_block(world):
    os.environ["CCP_GS_MASKED"] = "0"
    v, col, rowp, colour, n = system()
    cuts = cuts_for(n, world)
    ghosts = halo_sizes(col, rowp, cuts)
    say(stage="partition", world=world, rows_per_block=[cuts[k + 1] - cuts[k] for k in range(world)], ghosts_per_block=ghosts,
        halo_bytes_per_sweep_per_block=[8 * g for g in ghosts])
    comm = capi.Comm(capi.comm_unique_id(), 0, 1, 0)
    xt = synth.x_true(n, 1234)
    for what, lo, hi in (("whole matrix as the one block of a communicator", 0, n), (f"block {world // 2} of {world}, couplings to the other blocks cut", cuts[world // 2], cuts[world // 2 + 1])):
        a, b_ = rowp[lo], rowp[hi]
        c = col[a:b_].astype(np.int64) - lo
        keep = (c >= 0) & (c < hi - lo)
        nnz = np.add.reduceat(keep.astype(np.int64), (rowp[lo:hi] - a)) if hi > lo else np.zeros(0, np.int64)
        vals, cols = v[a:b_][keep], c[keep].astype(np.int32)
        begin = np.zeros(hi - lo, dtype=np.int64)
        begin[1:] = np.cumsum(nnz[:-1])
        m = capi.CsrMatrix()
        t = time.time()
        m.upload_rows(comm, 0, hi - lo, vals, cols, begin.astype(np.int32), nnz.astype(np.int32), colour[lo:hi], 2)
        t_up = time.time() - t
        bvec = synth.csr_apply(vals, cols, np.concatenate([begin, [len(vals)]]).astype(np.int64), xt[lo:hi])
        m.gauss_seidel(bvec, 0.0, 2, check_every=0)                  # images built
        best = None
        for _ in range(3):
            _, rep = m.gauss_seidel(bvec, 0.0, SWEEPS, check_every=0)
            best = rep.seconds if best is None else min(best, rep.seconds)
        entries = int(len(vals))
        model = 12.0 * entries + 32.0 * (hi - lo)                    # bytes per sweep on the CSR path
        say(stage="sweep", what=what, rows=hi - lo, entries=entries, sweeps=SWEEPS, seconds=best, ms_per_sweep=1e3 * best / SWEEPS,
            row_updates_per_s=(hi - lo) * SWEEPS / best, frac_of_8TBps=model * SWEEPS / best / 8e12, upload_rows_seconds=round(t_up, 3), path=m.last_path())
        m.close()
    comm.close()


def mode_threads(world):
    if not os.environ.get("CCP_GS_RCCL_LIB"):
        raise SystemExit("CCP_GS_RCCL_LIB must name tests/cpp/libfake_rccl.so")
    os.environ["CCP_GS_MASKED"] = "0"
    v, col, rowp, colour, n = system()
    cuts = cuts_for(n, world)
    xt = synth.x_true(n, 1234)
    b = synth.csr_apply(v, col, rowp, xt)
    iters = 10
    one = capi.CsrMatrix().upload_compressed(v, col, rowp.astype(np.int32)).set_colouring(colour, 2)
    want, rep1 = one.gauss_seidel(b, 0.0, iters, check_every=0)
    one.close()
    say(stage="one GPU", sweeps=iters, seconds=rep1.seconds)
    uid = capi.comm_unique_id()
    out, err = [None] * world, [None] * world

    def body(rank):
        try:
            comm = capi.Comm(uid, rank, world, 0)
            lo, hi = cuts[rank], cuts[rank + 1]
            a, e = rowp[lo], rowp[hi]
            m = capi.CsrMatrix()
            t = time.time()
            m.upload_rows(comm, lo, n, v[a:e], col[a:e], (rowp[lo:hi] - a).astype(np.int32), np.diff(rowp[lo:hi + 1]).astype(np.int32), colour[lo:hi], 2)
            t_up = time.time() - t
            x, rep = m.gauss_seidel(b[lo:hi], 0.0, iters, check_every=0)
            out[rank] = (x, rep.seconds, m.rows_info(), t_up)
            m.close()
            comm.close()
        except Exception as ex:  # noqa: BLE001
            err[rank] = repr(ex)

    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(600)
    if any(err) or any(o is None for o in out):
        say(stage="threads", ok=False, error=err)
        return
    got = np.concatenate([o[0] for o in out])
    say(stage="threads", ok=True, world=world, sweeps=iters, bit_identical_to_one_gpu=bool(np.array_equal(got, want)),
        ghosts=[o[2]["n_ghost"] for o in out], peers=[o[2]["n_peers"] for o in out], edge_slices=[o[2]["edge_slices"] for o in out],
        values_sent_per_sweep=[o[2]["values_sent"] // iters for o in out], exchanges=[o[2]["exchanges"] for o in out],
        upload_rows_seconds=[round(o[3], 2) for o in out], seconds_all_blocks_sharing_one_card=max(o[1] for o in out))


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "block"
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    {"block": mode_block, "threads": mode_threads}[mode](world)
