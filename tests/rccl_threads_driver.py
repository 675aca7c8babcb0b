"""Child process of tests/test_gpu_rccl_multirank.py (never imported by the product): the library's OWN multi-rank
row-block path — ccp_comm_create, ccp_grid_attach_comm, ccp_grid_exchange_halos, ccp_grid_sweep_rowblocked,
ccp_grid_gauss_seidel_rowblocked, ccp_grid_residual_norm2_global — at world size 2..4 on ONE card.

The ranks are THREADS of this process; CCP_GS_RCCL_LIB points libccp_gs.so at tests/cpp/libfake_rccl.so, whose
ncclSend/ncclRecv are matched device-to-device copies on the callers' streams (see that file).  Everything
above the twelve nccl* symbols is the production code path: the grouped send/recv pairs of issue_exchange(), the
in-launch edge hand-off that lets the messages leave beside the rest of the pass, the all-gathered partition
check, the all-reduced stop rule.

usage: rccl_threads_driver.py '<json list of cases>'   ->  one JSON line per case on stdout
"""
import ctypes as C
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from coursecomputationalphotography_amd import capi, rowblock, synth  # noqa: E402


def transport_stats():
    lib = C.CDLL(os.environ["CCP_GS_RCCL_LIB"])
    v = [C.c_long() for _ in range(5)]
    lib.fake_rccl_stats(*[C.byref(x) for x in v])
    return dict(zip(("sends", "recvs", "bytes", "allreduces", "allgathers"), [x.value for x in v]))


def run_ranks(world, fn):
    """fn(rank, comm) on `world` threads; returns the per-rank results, re-raising the first failure."""
    uid = capi.comm_unique_id()
    out, err = [None] * world, [None] * world

    def body(rank):
        comm = None
        try:
            comm = capi.Comm(uid, rank, world, 0)
            out[rank] = fn(rank, comm)
        except Exception as e:  # noqa: BLE001 - reported to the parent
            err[rank] = e
        finally:
            if comm is not None:
                comm.close()

    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    if any(t.is_alive() for t in ts):
        raise RuntimeError("a rank thread did not finish")
    return out, err


def system(g, scale=None, base=None):
    if base is None:
        g.randomize_x(1234, 0.0, 255.0)      # x_true: a function of (seed, x, y) only, so every block sees the same image
        g.b_from_x()
    else:
        for ch, s in enumerate(scale):
            lo = g.first_local_row
            g.set_b(np.ascontiguousarray(base[lo:lo + g.local_rows]) * s, ch)
    g.fill_x(1.0)


def case_sweep(c):
    """Owned rows after `iters` sweeps == the one-block sweep, bit for bit; the global residual too."""
    W, H, C_, world, ghost, iters, overlap = c["W"], c["H"], c.get("C", 1), c["world"], c["ghost"], c["iters"], c["overlap"]
    whole = capi.Grid(W, H, C_)
    system(whole)
    whole.sweep(iters)
    want = np.stack([whole.get_x(ch) for ch in range(C_)])
    rr_w, bb_w = whole.residual_norm2()
    whole.close()
    parts = rowblock.partition_rows(H, world)
    before = transport_stats()

    def rank_fn(rank, comm):
        assert comm.info()["rccl_version"] == 99901, "not the test transport"
        rb, rc = parts[rank]
        g = capi.Grid(W, H, C_, rb, rc, ghost, 0)
        system(g)
        g.attach_comm(comm)
        g.set_overlap(overlap)
        g.exchange_halos()
        done = 0
        for n in c.get("calls", [iters]):                     # several calls: intervals that straddle calls
            g.sweep_rowblocked(n)
            done += n
        assert done == iters
        rr, bb = g.residual_norm2_global()
        owned = np.stack([g.get_x_owned(ch) for ch in range(C_)])
        stats = g.comm_stats()
        g.attach_comm(None)
        g.close()
        return owned, rr, bb, stats

    out, err = run_ranks(world, rank_fn)
    if any(err):
        return {"ok": False, "error": [repr(e) for e in err]}
    got = np.concatenate([o[0] for o in out], axis=1)
    after = transport_stats()
    rel = float(np.linalg.norm(got - want) / np.linalg.norm(want))
    # the residual is summed block by block (then over ranks): equal up to the order of the additions
    rr = out[0][1]
    return {"ok": True, "bit_identical": bool(np.array_equal(got, want)), "rel_l2": rel,
            "residual_close": bool(np.allclose(rr, rr_w, rtol=1e-12, atol=0.0)),
            "residual_same_on_all_ranks": bool(all(np.array_equal(o[1], rr) for o in out)),
            "exchanges": [int(o[3][0]) for o in out], "wait_mode": [int(o[3][1]) for o in out],
            "sends": after["sends"] - before["sends"], "recvs": after["recvs"] - before["recvs"],
            "bytes": after["bytes"] - before["bytes"]}


def case_stop_rule(c):
    """`while (eps > epsilon && cnt < max_iteration)` (sparse-matrix.h:356,376) with the step all-reduced over the
    blocks: the same stop sweep as the one-block solve and as the CPU oracle; the last channel's iterate identical."""
    W, H, world, ghost, eps = c["W"], c["H"], c["world"], c["ghost"], c["eps"]
    scale = c.get("scale", [1e-3, 3e-4])
    base = synth.poisson_system(W, H, 1234)[0].reshape(H, W)
    ref = capi.Grid(W, H, len(scale))
    system(ref, scale, base)
    reps_w = ref.gauss_seidel(eps, 600, 1)
    want = [ref.get_x(ch) for ch in range(len(scale))]
    ref.close()
    parts = rowblock.partition_rows(H, world)

    def rank_fn(rank, comm):
        rb, rc = parts[rank]
        g = capi.Grid(W, H, len(scale), rb, rc, ghost, 0)
        system(g, scale, base)
        g.attach_comm(comm)
        g.exchange_halos()
        reps = g.gauss_seidel_rowblocked(eps, 600, 1)
        owned = [g.get_x_owned(ch) for ch in range(len(scale))]
        res = [(r.converged, r.iterations, r.last_l1_step) for r in reps]
        g.attach_comm(None)
        g.close()
        return owned, res

    out, err = run_ranks(world, rank_fn)
    if any(err):
        return {"ok": False, "error": [repr(e) for e in err]}
    its_w = [r.iterations for r in reps_w]
    last = int(np.argmax(its_w))
    got_last = np.concatenate([o[0][last] for o in out])
    # the oracle's stop sweep of the same rule on the colour-ordered matrix (one channel at a time)
    import oracle
    v, col, r = synth.poisson_csr(W, H)
    colour = oracle.grid_colour(W, H)
    orc = oracle.Oracle()
    its_o = [orc.multicolour_gauss_seidel(v, col, r, colour, (base * s).ravel(), eps, 600)[1] for s in scale]
    return {"ok": True, "iterations_one_block": its_w, "iterations_oracle": [int(k) for k in its_o],
            "iterations_ranks": [[t[1] for t in o[1]] for o in out], "converged": [[t[0] for t in o[1]] for o in out],
            "step_ranks": [[t[2] for t in o[1]] for o in out], "step_one_block": [r.last_l1_step for r in reps_w],
            "last_channel_bit_identical": bool(np.array_equal(got_last, want[last]))}


def case_bad_partition(c):
    """A partition that is not the rank-ordered contiguous row blocks of one image is refused on EVERY rank."""
    W, H, world, ghost = c["W"], c["H"], c["world"], c["ghost"]
    parts = c["parts"]

    def rank_fn(rank, comm):
        rb, rc = parts[rank]
        g = capi.Grid(W, H, 1, rb, rc, ghost, 0)
        try:
            g.attach_comm(comm)
            st = 0
        except capi.CcpError as e:
            st = e.status
        g.close()
        return st

    out, err = run_ranks(world, rank_fn)
    if any(err):
        return {"ok": False, "error": [repr(e) for e in err]}
    return {"ok": True, "status": out}


def case_late_flag(c):
    """the polling wait on the edge flag (wait mode 1) used to give up after 2 s and let the exchange
    send rows that were not final.  Forced here: the flag is only published after the pass
    (CCP_GS_EDGE_SIGNAL=0) and the polling kernel gives up at once (CCP_GS_EDGE_TIMEOUT_TICKS=1).  Every rank must
    then either FAIL (CCP_ERR_STATE when results are handed to the host) or hold the right rows — never wrong
    rows without an error."""
    os.environ.update({"CCP_GS_EDGE_WAIT": "spin", "CCP_GS_EDGE_SIGNAL": "0", "CCP_GS_EDGE_TIMEOUT_TICKS": "1"})
    W, H, world, ghost, iters = c["W"], c["H"], c["world"], c["ghost"], c["iters"]
    whole = capi.Grid(W, H, 1)
    system(whole)
    whole.sweep(iters)
    want = whole.get_x()
    whole.close()
    parts = rowblock.partition_rows(H, world)

    def rank_fn(rank, comm):
        rb, rc = parts[rank]
        g = capi.Grid(W, H, 1, rb, rc, ghost, 0)
        status, owned, mode = 0, None, None
        try:
            system(g)
            g.attach_comm(comm)
            mode = g.comm_stats()[1]
            g.set_overlap(True)
            g.exchange_halos()
            g.sweep_rowblocked(iters)
            g.synchronize()
            owned = g.get_x_owned()
        except capi.CcpError as e:
            status = e.status
        g.close()
        return status, owned, mode

    out, err = run_ranks(world, rank_fn)
    for k in ("CCP_GS_EDGE_WAIT", "CCP_GS_EDGE_SIGNAL", "CCP_GS_EDGE_TIMEOUT_TICKS"):
        os.environ.pop(k, None)
    if any(err):
        return {"ok": False, "error": [repr(e) for e in err]}
    verdict = []
    for rank, (status, owned, mode) in enumerate(out):
        rb, rc = parts[rank]
        right = owned is not None and bool(np.array_equal(owned, want[rb:rb + rc]))
        verdict.append({"status": status, "rows_right": right, "wait_mode": mode})
    return {"ok": True, "ranks": verdict}


CASES = {"late_flag": case_late_flag, "sweep": case_sweep, "stop_rule": case_stop_rule, "bad_partition": case_bad_partition}


def main():
    if not os.environ.get("CCP_GS_RCCL_LIB"):
        raise SystemExit("CCP_GS_RCCL_LIB must name the test transport")
    for c in json.loads(sys.argv[1]):
        try:
            res = CASES[c["kind"]](c)
        except Exception as e:  # noqa: BLE001
            res = {"ok": False, "error": repr(e)}
        print(json.dumps({"case": c, **res}), flush=True)


if __name__ == "__main__":
    main()
