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


def region_mask(c):
    if not c.get("mask"):
        return None
    return synth.disc_mask(c["W"], c["H"], seed=c.get("seed", 4321), n_discs=c.get("discs", 40), rmin=300.0, rmax=1400.0).astype(np.uint8)


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
    mask = region_mask(c)                                  # a Dirichlet-mask grid (the region of BASELINE configs[4]) or None
    whole = capi.Grid(W, H, C_, mask=mask)
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
        g = capi.Grid(W, H, C_, rb, rc, ghost, 0, mask=mask)
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
    mask = region_mask(c)
    ref = capi.Grid(W, H, len(scale), mask=mask)
    system(ref, scale, base)
    reps_w = ref.gauss_seidel(eps, 600, 1)
    want = [ref.get_x(ch) for ch in range(len(scale))]
    ref.close()
    parts = rowblock.partition_rows(H, world)

    def rank_fn(rank, comm):
        rb, rc = parts[rank]
        g = capi.Grid(W, H, len(scale), rb, rc, ghost, 0, mask=mask)
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
    orc = oracle.Oracle()
    if mask is None:
        v, col, r = synth.poisson_csr(W, H)
        colour = oracle.grid_colour(W, H)
        its_o = [orc.multicolour_gauss_seidel(v, col, r, colour, (base * s).ravel(), eps, 600)[1] for s in scale]
    else:                                                  # the region's own matrix, unknowns in raster order
        v, col, r, colour, ys, xs = synth.masked_laplacian_csr(mask != 0)
        its_o = [orc.multicolour_gauss_seidel(v, col, r, colour, (base * s)[ys, xs], eps, 600)[1] for s in scale]
    return {"ok": True, "iterations_one_block": its_w, "iterations_oracle": [int(k) for k in its_o],
            "iterations_ranks": [[t[1] for t in o[1]] for o in out], "converged": [[t[0] for t in o[1]] for o in out],
            "step_ranks": [[t[2] for t in o[1]] for o in out], "step_one_block": [r.last_l1_step for r in reps_w],
            "last_channel_bit_identical": bool(np.array_equal(got_last, want[last])),
            "all_channels_bit_identical": bool(all(np.array_equal(np.concatenate([o[0][ch] for o in out]), want[ch]) for ch in range(len(scale))))}


def case_grid_cg(c):
    """conjugateGradient on row blocks of the grid (plain or Dirichlet-mask): iterates equal the one-block loop's to rounding,
    the same stop iteration on every rank."""
    W, H, C_, world, ghost = c["W"], c["H"], c.get("C", 1), c["world"], c["ghost"]
    eps, iters = c["eps"], c["iters"]
    mask = region_mask(c)
    os.environ["CCP_GS_CG_FUSED"] = "0"                    # the one-block comparison on the same three-vector loop
    whole = capi.Grid(W, H, C_, mask=mask)
    system(whole)
    whole.fill_x(0.0)
    reps_w = whole.conjugate_gradient(eps, iters)
    want = np.stack([whole.get_x(ch) for ch in range(C_)])
    whole.close()
    os.environ.pop("CCP_GS_CG_FUSED", None)
    parts = rowblock.partition_rows(H, world)

    def rank_fn(rank, comm):
        rb, rc = parts[rank]
        g = capi.Grid(W, H, C_, rb, rc, ghost, 0, mask=mask)
        system(g)
        g.fill_x(0.0)
        g.attach_comm(comm)
        reps = g.conjugate_gradient_rowblocked(eps, iters)
        owned = np.stack([g.get_x_owned(ch) for ch in range(C_)])
        # a sweep after the solve must refresh the ghost rows by itself
        g.sweep_rowblocked(2)
        after = np.stack([g.get_x_owned(ch) for ch in range(C_)])
        res = [(r.converged, r.iterations, r.last_l1_step) for r in reps]
        g.attach_comm(None)
        g.close()
        return owned, res, after

    out, err = run_ranks(world, rank_fn)
    if any(err):
        return {"ok": False, "error": [repr(e) for e in err]}
    got = np.concatenate([o[0] for o in out], axis=1)
    after = np.concatenate([o[2] for o in out], axis=1)
    ref = capi.Grid(W, H, C_, mask=mask)
    system(ref)
    for ch in range(C_):
        ref.set_x(got[ch], ch)
    ref.sweep(2)
    after_w = np.stack([ref.get_x(ch) for ch in range(C_)])
    ref.close()
    return {"ok": True, "rel_diff": float(np.linalg.norm(got - want) / np.linalg.norm(want)),
            "iterations_one_block": [r.iterations for r in reps_w], "converged_one_block": [r.converged for r in reps_w],
            "iterations_ranks": [[t[1] for t in o[1]] for o in out], "converged_ranks": [[t[0] for t in o[1]] for o in out],
            "rnorm_ranks": [[t[2] for t in o[1]] for o in out], "rnorm_one_block": [r.last_l1_step for r in reps_w],
            "sweep_after_solve_bit_identical": bool(np.array_equal(after, after_w))}


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


# ---- row blocks of a general CSR matrix (ccp_csr_upload_rows; SURVEY §8e "row-block by unknown index with a halo index list")

def csr_matrix(c):
    """(values, col, row_offset, colour, n_colours) of the whole matrix of a case."""
    if c["matrix"] == "mask":                      # BASELINE configs[4] in small: discs + brush trail, Dirichlet Laplacian
        mask = synth.disc_mask(c["W"], c["H"], seed=c.get("seed", 4321), n_discs=c.get("discs", 24), rmin=300.0, rmax=c.get("rmax", 1400.0))
        v, col, rowp, colour, _, _ = synth.masked_laplacian_csr(mask)
        return v, col, rowp, colour, 2
    # random symmetric pattern, diagonally dominant; couplings reach far across the blocks (every block talks to every other)
    n, g = c["n"], synth.rng(c.get("seed", 7))
    pairs = set()
    for i in range(n):
        for j in g.integers(0, n, c.get("deg", 3)):
            j = int(j)
            if j != i:
                pairs.add((i, j))
                pairs.add((j, i))
    rows = [[] for _ in range(n)]
    for i, j in pairs:
        rows[i].append(j)
    vals, cols, rowp = [], [], [0]
    for i in range(n):
        js = sorted(rows[i] + [i])
        w = {j: -g.uniform(0.2, 1.0) for j in js if j != i}
        w[i] = 0.0 if c.get("empty_rows") and i % 97 == 5 else 1.0 + sum(-t for t in w.values())
        for j in js:
            if w[j] != 0.0:
                cols.append(j)
                vals.append(w[j])
        rowp.append(len(cols))
    v, col, rowp = np.array(vals), np.array(cols, dtype=np.int32), np.array(rowp, dtype=np.int32)
    whole = capi.CsrMatrix().upload_compressed(v, col, rowp)
    colour, nc = whole.get_colouring()             # the library's greedy colouring of the whole matrix
    whole.close()
    return v, col, rowp, colour.copy(), nc


def slack_rows(v, col, rowp, lo, hi, slack):
    """The reference's five arrays for rows [lo, hi): `slack` unused entries after every row (sparse-matrix.h:670-676)."""
    nnz = np.diff(rowp[lo:hi + 1]).astype(np.int32)
    begin = np.zeros(hi - lo, dtype=np.int32)
    if hi > lo:
        begin[1:] = np.cumsum(nnz[:-1] + slack)
    total = int(begin[-1] + nnz[-1] + slack) if hi > lo else 0
    values = np.full(total, np.nan)                # slack is never read
    cols = np.full(total, -7, dtype=np.int32)
    for i in range(hi - lo):
        a = rowp[lo + i]
        values[begin[i]:begin[i] + nnz[i]] = v[a:a + nnz[i]]
        cols[begin[i]:begin[i] + nnz[i]] = col[a:a + nnz[i]]
    return values, cols, begin, nnz


def case_csr_rows(c):
    """Blocks of rows of one matrix on `world` ranks: the sweep's iterates, b := A x and the stop sweep are the one-GPU
    handle's (and the CPU oracle's on P A P^T), bit for bit."""
    import oracle
    v, col, rowp, colour, nc = csr_matrix(c)
    n = len(rowp) - 1
    world, iters, eps = c["world"], c["iters"], c.get("eps", 0.0)
    xt = synth.x_true(n, 1234)
    b = synth.csr_apply(v, col, rowp, xt)
    x0 = synth.x_true(n, 99) if c.get("x0") else None
    os.environ["CCP_GS_MASKED"] = "0"              # the one-GPU comparison on the stored matrix too
    os.environ["CCP_GS_ROWS_OVERLAP"] = "1" if c.get("overlap", True) else "0"
    whole = capi.CsrMatrix().upload_compressed(v, col, rowp).set_colouring(colour, nc)
    want, rep_w = whole.gauss_seidel(b, eps, iters, x0, check_every=1 if eps > 0 else 0)
    y_w = whole.apply_to_vector(xt)
    rr_w, bb_w = whole.residual_norm2(b, want)
    cg = c.get("cg")                                  # [epsilon, max_iteration]: conjugateGradient (sparse-matrix.h:396-434) as well
    if cg:
        cg_w, cg_rep_w = whole.conjugate_gradient(b, cg[0], cg[1], init=x0)
        pcg_w, pcg_rep_w = whole.conjugate_gradient_jacobi(b, cg[0], cg[1])       # conjugateGradientEigen (:494-535), from x0 = 0
    whole.close()
    orc = oracle.Oracle()
    want_o, it_o, _ = orc.multicolour_gauss_seidel(v, col, rowp, colour, b, eps, iters, x0=x0)
    cuts = c.get("cuts") or [round(n * k / world) for k in range(world + 1)]
    cuts = [int(round(t * n)) if isinstance(t, float) else t for t in cuts]
    before = transport_stats()

    def rank_fn(rank, comm):
        assert comm.info()["rccl_version"] == 99901, "not the test transport"
        lo, hi = cuts[rank], cuts[rank + 1]
        values, cols, begin, nnz = slack_rows(v, col, rowp, lo, hi, c.get("slack", 0))
        m = capi.CsrMatrix()
        m.upload_rows(comm, lo, n, values, cols, begin, nnz, colour[lo:hi], nc)
        x, rep = m.gauss_seidel(b[lo:hi], eps, iters, None if x0 is None else x0[lo:hi], check_every=1 if eps > 0 else 0)
        y = m.apply_to_vector(xt[lo:hi])
        rr, bb = m.residual_norm2(b[lo:hi], x)
        own_colour, own_nc = m.get_colouring()
        unsupported = []
        cg_out = None
        if cg:
            xc, rc = m.conjugate_gradient(b[lo:hi], cg[0], cg[1], init=None if x0 is None else x0[lo:hi])
            xp, rp = m.conjugate_gradient_jacobi(b[lo:hi], cg[0], cg[1])
            cg_out = (xc, rc.converged, rc.iterations, rc.last_l1_step, xp, rp.converged, rp.iterations)
        info = m.rows_info()
        for what, fn in (("lexicographic", lambda: m.gauss_seidel(b[lo:hi], 0.0, 1, ordering=capi.ORDER_LEXICOGRAPHIC)),
                         ("insert", lambda: m.insert(1.0, 0, 0))):
            try:
                fn()
                unsupported.append((what, 0))
            except capi.CcpError as e:
                unsupported.append((what, e.status))
        path = m.last_path()
        m.close()
        return x, (rep.converged, rep.iterations, rep.last_l1_step), y, rr, bb, info, bool(np.array_equal(own_colour, colour[lo:hi]) and own_nc == nc), unsupported, path, cg_out

    out, err = run_ranks(world, rank_fn)
    os.environ.pop("CCP_GS_MASKED", None)
    os.environ.pop("CCP_GS_ROWS_OVERLAP", None)
    if any(err):
        return {"ok": False, "error": [repr(e) for e in err]}
    after = transport_stats()
    got = np.concatenate([o[0] for o in out])
    y = np.concatenate([o[2] for o in out])
    ghosts = [o[5]["n_ghost"] for o in out]
    cg_res = {}
    if cg:
        xc = np.concatenate([o[9][0] for o in out])
        cg_res = {"cg_rel_diff": float(np.linalg.norm(xc - cg_w) / np.linalg.norm(cg_w)), "cg_iterations_one_gpu": cg_rep_w.iterations,
                  "cg_converged_one_gpu": cg_rep_w.converged, "cg_iterations_ranks": [o[9][2] for o in out], "cg_converged_ranks": [o[9][1] for o in out],
                  "cg_rnorm_ranks": [o[9][3] for o in out], "cg_rnorm_one_gpu": cg_rep_w.last_l1_step,
                  "pcg_rel_diff": float(np.linalg.norm(np.concatenate([o[9][4] for o in out]) - pcg_w) / np.linalg.norm(pcg_w)),
                  "pcg_iterations_one_gpu": pcg_rep_w.iterations, "pcg_converged_one_gpu": pcg_rep_w.converged,
                  "pcg_iterations_ranks": [o[9][6] for o in out], "pcg_converged_ranks": [o[9][5] for o in out]}
    return {"ok": True, "n": n, "colours": nc, "cuts": cuts, **cg_res,
            "bit_identical_to_one_gpu": bool(np.array_equal(got, want)),
            "bit_identical_to_oracle": bool(np.array_equal(got, want_o)),
            "spmv_bit_identical": bool(np.array_equal(y, y_w)),
            "residual_close": bool(np.allclose([out[0][3], out[0][4]], [rr_w, bb_w], rtol=1e-12, atol=0.0)),
            "residual_same_on_all_ranks": bool(all(o[3] == out[0][3] and o[4] == out[0][4] for o in out)),
            "iterations_one_gpu": rep_w.iterations, "iterations_oracle": int(it_o), "iterations_ranks": [o[1][1] for o in out],
            "converged_ranks": [o[1][0] for o in out], "converged_one_gpu": rep_w.converged,
            "step_ranks": [o[1][2] for o in out], "step_one_gpu": rep_w.last_l1_step,
            "ghosts": ghosts, "peers": [o[5]["n_peers"] for o in out], "edge_slices": [o[5]["edge_slices"] for o in out], "values_sent": [o[5]["values_sent"] for o in out],
            "exchanges": [o[5]["exchanges"] for o in out], "own_colours_ok": all(o[6] for o in out),
            "unsupported": [o[7] for o in out], "path": [o[8] for o in out],
            "sends": after["sends"] - before["sends"], "recvs": after["recvs"] - before["recvs"], "bytes": after["bytes"] - before["bytes"]}


def case_csr_rows_refused(c):
    """One rank's arguments are wrong (a colouring with two coupled rows of one colour; a block that leaves a gap; a
    column outside the matrix): EVERY rank returns an error and nobody hangs."""
    v, col, rowp, colour, nc = csr_matrix(c)
    n = len(rowp) - 1
    world, fault = c["world"], c["fault"]
    cuts = [round(n * k / world) for k in range(world + 1)]

    def rank_fn(rank, comm):
        lo, hi = cuts[rank], cuts[rank + 1]
        values, cols, begin, nnz = slack_rows(v, col, rowp, lo, hi, 0)
        colr = colour[lo:hi].copy()
        if rank == c["rank"]:
            if fault == "colouring":
                colr[:] = 0
            elif fault == "gap":
                lo += 1
                values, cols, begin, nnz = slack_rows(v, col, rowp, lo, hi, 0)
                colr = colour[lo:hi].copy()
            elif fault == "column":
                cols = cols.copy()
                cols[len(cols) // 2] = n + 5
        m = capi.CsrMatrix()
        try:
            m.upload_rows(comm, lo, n, values, cols, begin, nnz, colr, nc)
            st = 0
        except capi.CcpError as e:
            st = e.status
        after = None
        try:
            x1, _ = m.gauss_seidel(np.zeros(hi - lo), 0.0, 1)
            if fault == "none":
                # the handle takes another block set-up (same arguments) and gives the same iterate; its counters start again
                sent1 = m.rows_info()["values_sent"]
                m.upload_rows(comm, lo, n, values, cols, begin, nnz, colr, nc)
                if m.rows_info()["values_sent"] != 0:
                    after = -1
                x2, _ = m.gauss_seidel(np.zeros(hi - lo), 0.0, 1)
                if not np.array_equal(x1, x2) or m.rows_info()["values_sent"] != sent1:
                    after = -2
        except capi.CcpError as e:
            after = e.status
        m.close()
        return st, after

    out, err = run_ranks(world, rank_fn)
    if any(err):
        return {"ok": False, "error": [repr(e) for e in err]}
    return {"ok": True, "status": [o[0] for o in out], "solve_after": [o[1] for o in out]}


CASES = {"grid_cg": case_grid_cg, "csr_rows": case_csr_rows, "csr_rows_refused": case_csr_rows_refused, "late_flag": case_late_flag, "sweep": case_sweep, "stop_rule": case_stop_rule, "bad_partition": case_bad_partition}


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
