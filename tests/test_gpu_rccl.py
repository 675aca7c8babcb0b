"""GPU: the RCCL path behind the C ABI (ccp_comm_*, ccp_grid_*_rowblocked) at world size 1 — the whole
code path a multi-GPU run takes (communicator init, partition all-gather, zero-neighbour exchange,
all-reduced stop rule and residual, teardown) on the one GPU a test box has — from Python and from the
C++ host tests/cpp/rowblock_driver.cpp; plus the in-launch edge hand-off that lets the exchange overlap
the sweep, exercised with several row blocks on ONE card (halo messages staged by the test)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def capi():
    from coursecomputationalphotography_amd import capi
    assert capi.device_count() >= 1
    return capi


def make(capi, W, H, C=1, **kw):
    g = capi.Grid(W, H, C, **kw)
    g.randomize_x(1234, 0.0, 255.0)
    g.b_from_x()
    g.fill_x(1.0)
    return g


def test_world1_full_rccl_path(capi):
    comm = capi.Comm(capi.comm_unique_id(), 0, 1, 0)
    info = comm.info()
    assert info["rank"] == 0 and info["world"] == 1 and info["rccl_version"] >= 20000
    assert np.array_equal(comm.all_reduce_sum([1.5, -2.0, 3.25]), [1.5, -2.0, 3.25])
    assert np.array_equal(comm.all_reduce_max([7.0]), [7.0])
    W, H, C = 1500, 1100, 2
    ref = make(capi, W, H, C)
    ref.sweep(21)
    want = [ref.get_x(ch) for ch in range(C)]
    rr_w, bb_w = ref.residual_norm2()
    g = make(capi, W, H, C)
    g.attach_comm(comm)                                   # all-gather of the (one-block) partition
    assert g.comm_stats()[1] == -1                        # no neighbours
    g.exchange_halos()                                    # zero-neighbour exchange
    g.sweep_rowblocked(21)
    for ch in range(C):
        assert np.array_equal(g.get_x(ch), want[ch])
    rr, bb = g.residual_norm2_global()                    # all-reduce over one rank
    assert np.array_equal(rr, rr_w) and np.array_equal(bb, bb_w)
    g.attach_comm(None)
    g.close()
    ref.close()
    # the reference loop with its stop rule (sparse-matrix.h:356), step all-reduced: same stop sweep and
    # iterate as the one-GPU entry point (small right-hand sides so that the L1 step falls below eps = 10's scale)
    from coursecomputationalphotography_amd import synth
    W, H = 96, 80
    base = synth.poisson_system(W, H, 1234)[0]
    ref, g = capi.Grid(W, H, 2), capi.Grid(W, H, 2)
    for h in (ref, g):
        h.set_b(base * 1e-3, 0)
        h.set_b(base * 3e-4, 1)
        h.fill_x(1.0)
    g.attach_comm(comm)
    eps = 0.5
    reps_w = ref.gauss_seidel(eps, 600, 1)
    reps = g.gauss_seidel_rowblocked(eps, 600, 1)
    assert all(r.converged == 1 for r in reps_w) and all(r.converged == 1 for r in reps)
    assert [r.iterations for r in reps] == [r.iterations for r in reps_w]
    # all channels run until the last one stops (documented): the channel that stops last is bit-identical
    last = int(np.argmax([r.iterations for r in reps_w]))
    assert np.array_equal(g.get_x(last), ref.get_x(last))
    assert abs(reps[last].last_l1_step - reps_w[last].last_l1_step) <= 1e-10 * reps_w[last].last_l1_step
    # fixed count, no rule: identical to the plain sweep
    for h in (ref, g):
        h.fill_x(1.0)
    ref.sweep(33)
    rep = g.gauss_seidel_rowblocked(0.0, 33, 0)[0]
    assert rep.iterations == 33 and rep.converged == 0
    assert np.array_equal(g.get_x(0), ref.get_x(0)) and np.array_equal(g.get_x(1), ref.get_x(1))
    g.attach_comm(None)
    g.close()
    ref.close()
    comm.close()


def test_attach_rejects_a_partition_that_is_not_the_image(capi):
    comm = capi.Comm(capi.comm_unique_id(), 0, 1, 0)
    g = capi.Grid(256, 256, 1, 0, 128, 0, 0)             # half the image as the only rank of the communicator
    with pytest.raises(capi.CcpError) as e:
        g.attach_comm(comm)
    assert e.value.status == 1
    with pytest.raises(capi.CcpError) as e:
        g.sweep_rowblocked(2)                              # nothing attached
    assert e.value.status == 5
    g.close()
    comm.close()


def test_cpp_host_drives_the_row_blocked_path(capi, tmp_path):
    """tests/cpp/rowblock_driver.cpp: a C++ program that sees only include/ccp_gs.h."""
    cpp = os.path.join(ROOT, "tests", "cpp")
    subprocess.check_call(["make", "-C", cpp], stdout=subprocess.DEVNULL)
    W, H, iters = 2048, 1200, 37
    out = subprocess.run([os.path.join(cpp, "rowblock_driver"), "1", "0", str(tmp_path / "id.bin"), str(W), str(H), "16", str(iters)],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    tok = out.stdout.split()
    val = {tok[i]: tok[i + 1] for i in range(0, len(tok) - 1)}
    g = make(capi, W, H)
    g.sweep(iters)
    rr, bb = g.residual_norm2()
    assert int(val["iterations"]) == iters
    assert float(val["rr"]) == rr[0] and float(val["bb"]) == bb[0] and float(val["abs"]) == g.abs_sum()[0]
    g.close()


@pytest.mark.parametrize("env", [{}, {"CCP_GS_EDGE_WAIT": "spin"}, {"CCP_GS_EDGE_SIGNAL": "0"}])
@pytest.mark.parametrize("W,H,parts,ghost,iters", [(16384, 1536, 3, 32, 40), (1000, 300, 2, 8, 13)])
def test_edge_hand_off_inside_the_launch(capi, monkeypatch, env, W, H, parts, ghost, iters):
    """Row blocks of one image on ONE card.  Each interval's last pass finishes the rows the neighbours take
    first and publishes the edge flag from inside the launch; a side stream waits for the flag only
    (hipStreamWaitValue64, or the polling kernel) and copies those rows into the neighbours' ghost rows
    while the rest of the pass is still running.  Owned rows must equal the one-block sweep bit for bit —
    they would not if the flag fired before the edge rows were final."""
    import torch
    from coursecomputationalphotography_amd import rowblock
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    whole = make(capi, W, H)
    whole.sweep(iters)
    want = whole.get_x()
    whole.close()
    rows = rowblock.partition_rows(H, parts)
    blocks = []
    for rb, rc in rows:
        blk = rowblock.GridBlock(W, H, 1, rb, rc, ghost, 0)
        blk.grid.randomize_x(1234, 0.0, 255.0)
        blk.grid.b_from_x()
        blk.grid.fill_x(1.0)
        blocks.append(blk)
    ipe = ghost // 2
    side = [torch.cuda.Stream() for _ in blocks]

    def exchange(after_edges):
        # neighbour copies on each RECEIVER's side stream, which waits for the SENDER's edge flag
        for i, blk in enumerate(blocks):
            gt, gb, own_lo = blk.ghost_top, blk.ghost_bottom, blk.ghost_top
            own_hi = own_lo + blk.row_count
            for src, dst_rows, src_rows in ((i - 1, slice(0, gt), None), (i + 1, slice(own_hi, own_hi + gb), None)):
                if src < 0 or src >= len(blocks):
                    continue
                s = blocks[src]
                s_lo, s_hi = s.ghost_top, s.ghost_top + s.row_count
                take = s.x_rows[:, s_hi - gt:s_hi] if src == i - 1 else s.x_rows[:, s_lo:s_lo + gb]
                if after_edges:
                    s.grid.stream_wait_edges(side[i].cuda_stream)
                    blk.grid.stream_wait_edges(side[i].cuda_stream)      # and my own pass has started (its input is intact)
                else:
                    side[i].wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side[i]):
                    blk.x_rows[:, dst_rows].copy_(take, non_blocking=True)
        for i, blk in enumerate(blocks):
            torch.cuda.current_stream().wait_stream(side[i])
            blk.halo_refreshed()

    exchange(False)
    done = 0
    while done < iters:
        room = min(ipe, iters - done)
        if room == ipe:
            for blk in blocks:
                blk.sweep_edges_first(room, ghost)
            exchange(True)
        else:
            for blk in blocks:
                blk.sweep(room)
        done += room
    got = np.concatenate([blk.grid.get_x_owned() for blk in blocks])
    for blk in blocks:
        blk.close()
    assert np.array_equal(got, want)


def test_world1_csr_row_block_and_mask_grid_through_real_rccl(capi, monkeypatch):
    """ccp_csr_upload_rows with the process's real RCCL at world 1: the all-gathered partition, the (empty) halo lists,
    the all-reduced stop rule and residual run through librccl itself; results are the one-GPU handle's."""
    from coursecomputationalphotography_amd import synth
    monkeypatch.setenv("CCP_GS_MASKED", "0")
    comm = capi.Comm(capi.comm_unique_id(), 0, 1, 0)
    mask = synth.disc_mask(300, 220, n_discs=20, rmin=300.0, rmax=700.0)
    v, col, rowp, colour, ys, xs = synth.masked_laplacian_csr(mask)
    n = len(rowp) - 1
    xt = synth.x_true(n, 1234)
    b = synth.csr_apply(v, col, rowp, xt)
    one = capi.CsrMatrix().upload_compressed(v, col, rowp).set_colouring(colour, 2)
    blk = capi.CsrMatrix().upload_rows(comm, 0, n, v, col, rowp[:-1], np.diff(rowp), colour, 2)
    info = blk.rows_info()
    assert (info["n_rows"], info["n_ghost"], info["n_peers"], info["edge_slices"]) == (n, 0, 0, 0)
    want, _ = one.gauss_seidel(b, 0.0, 9, check_every=0)
    got, rep = blk.gauss_seidel(b, 0.0, 9, check_every=0)
    assert rep.iterations == 9 and np.array_equal(got, want)
    want, rep_w = one.gauss_seidel(b * 1e-3, 2.0, 900)
    got, rep = blk.gauss_seidel(b * 1e-3, 2.0, 900)
    assert rep_w.converged == 1 and (rep.converged, rep.iterations) == (1, rep_w.iterations) and np.array_equal(got, want)
    assert np.array_equal(blk.apply_to_vector(xt), one.apply_to_vector(xt))
    assert np.allclose(blk.residual_norm2(b, got), one.residual_norm2(b, want), rtol=1e-13, atol=0.0)
    blk.close()
    one.close()
    # a Dirichlet-mask grid as the one row block of a communicator
    m8 = mask.astype(np.uint8)
    ref = make(capi, 300, 220, 1, mask=m8)
    g = make(capi, 300, 220, 1, mask=m8)
    g.attach_comm(comm)
    g.exchange_halos()
    ref.sweep(17)
    g.sweep_rowblocked(17)
    assert np.array_equal(g.get_x(0), ref.get_x(0))
    # conjugate gradient through the real all-reduce (one rank): the one-block three-vector loop's iterates
    monkeypatch.setenv("CCP_GS_CG_FUSED", "0")
    for h in (ref, g):
        h.fill_x(0.0)
    want_rep = ref.conjugate_gradient(1e-30, 20)[0]
    got_rep = g.conjugate_gradient_rowblocked(1e-30, 20)[0]
    assert (got_rep.iterations, got_rep.converged) == (want_rep.iterations, want_rep.converged) == (20, 0)
    assert np.allclose(g.get_x(0), ref.get_x(0), rtol=1e-11, atol=1e-9)
    # ... and the fused 72-byte loop (default) on the one-rank communicator against the one-block fused loop
    monkeypatch.delenv("CCP_GS_CG_FUSED")
    for h in (ref, g):
        h.fill_x(0.0)
    want_rep = ref.conjugate_gradient(1e-30, 20)[0]
    got_rep = g.conjugate_gradient_rowblocked(1e-30, 20)[0]
    assert got_rep.iterations == want_rep.iterations == 20
    assert np.allclose(g.get_x(0), ref.get_x(0), rtol=1e-11, atol=1e-9)
    blk2 = capi.CsrMatrix().upload_rows(comm, 0, n, v, col, rowp[:-1], np.diff(rowp), colour, 2)
    one2 = capi.CsrMatrix().upload_compressed(v, col, rowp)
    xa, ra = blk2.conjugate_gradient(b, 1e-30, 20)
    xb, rb_ = one2.conjugate_gradient(b, 1e-30, 20)
    assert ra.iterations == rb_.iterations == 20 and np.allclose(xa, xb, rtol=1e-11, atol=1e-9)
    blk2.close()
    one2.close()
    g.attach_comm(None)
    g.close()
    ref.close()
    comm.close()


def test_a_plain_upload_after_a_row_block_returns_to_the_one_gpu_forms(capi):
    """ccp_csr_upload_rows switches the grid twins and the one-block solver off (they know nothing of ghosts); a later
    plain ccp_csr_upload on the SAME handle must be the one-GPU form ccp_gs.h promises again: the region twin for a
    region matrix, the Poisson twin for SolveChannel's matrix."""
    from coursecomputationalphotography_amd import synth
    comm = capi.Comm(capi.comm_unique_id(), 0, 1, 0)
    mask = synth.disc_mask(300, 220, n_discs=20, rmin=300.0, rmax=700.0)
    v, col, rowp, colour, ys, xs = synth.masked_laplacian_csr(mask)
    n = len(rowp) - 1
    b = synth.csr_apply(v, col, rowp, synth.x_true(n, 1234))
    m = capi.CsrMatrix().upload_rows(comm, 0, n, v, col, rowp[:-1], np.diff(rowp), colour, 2)
    blocked, _ = m.gauss_seidel(b, 0.0, 6, check_every=0)
    assert m.last_path() == "sliced ELL"
    m.upload_compressed(v, col, rowp).set_colouring(colour, 2)
    again, _ = m.gauss_seidel(b, 0.0, 6, check_every=0)
    assert m.last_path().startswith("region grid") and np.array_equal(again, blocked)
    pv, pc, pr = synth.poisson_csr(40, 30)
    pb, _ = synth.poisson_system(40, 30, 7)
    m.upload_rows(comm, 0, 1200, pv, pc, pr[:-1], np.diff(pr), ((np.arange(1200) % 40 + np.arange(1200) // 40) & 1).astype(np.int32), 2)
    blocked, _ = m.gauss_seidel(pb, 0.0, 6, check_every=0)
    assert m.last_path() == "sliced ELL"
    m.upload_compressed(pv, pc, pr)
    again, _ = m.gauss_seidel(pb, 0.0, 6, check_every=0)
    assert m.last_path().startswith("Poisson grid") and np.array_equal(again, blocked)
    m.close()
    comm.close()


def test_a_refused_row_block_leaves_no_matrix_behind(capi):
    """After ANY refusal of ccp_csr_upload_rows the handle holds no matrix — not the previous block either."""
    from coursecomputationalphotography_amd import synth
    comm = capi.Comm(capi.comm_unique_id(), 0, 1, 0)
    pv, pc, pr = synth.poisson_csr(20, 10)
    colour = ((np.arange(200) % 20 + np.arange(200) // 20) & 1).astype(np.int32)
    m = capi.CsrMatrix().upload_rows(comm, 0, 200, pv, pc, pr[:-1], np.diff(pr), colour, 2)
    assert m.rows_info()["n_rows"] == 200
    with pytest.raises(capi.CcpError):                        # a block that is not the whole matrix at world 1
        m.upload_rows(comm, 0, 300, pv, pc, pr[:-1], np.diff(pr), colour, 2)
    with pytest.raises(capi.CcpError):
        m.rows_info()
    with pytest.raises(capi.CcpError):
        m.gauss_seidel(np.ones(200), 0.0, 1, check_every=0)
    m.close()
    comm.close()


_EXIT_CHILD = r"""
import importlib.util, os, sys
sys.path.insert(0, {root!r})
if {by_path!r}:
    spec = importlib.util.find_spec("torch")
    os.environ["CCP_GS_RCCL_LIB"] = os.path.join(os.path.dirname(spec.origin), "lib", "librccl.so")
from coursecomputationalphotography_amd import capi
assert not {by_path!r} or "torch" not in sys.modules
comm = capi.Comm(capi.comm_unique_id(), 0, 1, 0)
assert comm.info()["world"] == 1
comm.close()
import torch
assert torch.cuda.is_available()
print("clean", flush=True)
"""


@pytest.mark.parametrize("by_path", [True, False])
def test_a_process_that_used_the_communicator_and_imports_torch_exits_cleanly(by_path):
    """Regression test of round 2's at-exit abort ("double free or corruption (!prev)", exit code -6): a process that
    used RCCL through the library and imported torch AFTERWARDS died in librocm_smi64's exit handlers because the
    library had opened RCCL with RTLD_GLOBAL (csrc/ccp_comm.hip: rccl_api).  by_path: torch's RCCL bound by hand before
    torch is imported — the order that aborted; otherwise the binding's default order (torch first)."""
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    out = subprocess.run([sys.executable, "-c", _EXIT_CHILD.format(root=root, by_path=by_path)], capture_output=True, text=True, timeout=300)
    assert "clean" in out.stdout, out.stderr[-2000:]
    assert out.returncode == 0, (out.returncode, out.stderr[-2000:])
