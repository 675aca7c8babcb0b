"""GPU: the library's OWN multi-rank row-block path at world size 2, 3 and 4 on the one card a test box has.

The ranks are threads of a child process (tests/rccl_threads_driver.py, tests/cpp/rowblock_driver.cpp with
rank = -1) and CCP_GS_RCCL_LIB points libccp_gs.so at the test-only transport tests/cpp/libfake_rccl.so,
whose twelve nccl* symbols execute matched sends/receives as device-to-device copies on the callers' streams.
What runs above those symbols is the production code: issue_exchange()'s grouped ncclSend/ncclRecv pairs
(csrc/ccp_grid.hip), the in-launch edge hand-off that lets the messages leave beside the rest of a pass,
ccp_grid_attach_comm's all-gathered partition check, the all-reduced stop rule of
ccp_grid_gauss_seidel_rowblocked (the reference loop, sparse-matrix.h:356,376).  A real multi-GPU run differs
only below the nccl* calls."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CPP = os.path.join(ROOT, "tests", "cpp")
FAKE = os.path.join(CPP, "libfake_rccl.so")


@pytest.fixture(scope="module")
def fake_env():
    subprocess.check_call(["make", "-C", CPP], stdout=subprocess.DEVNULL)
    env = dict(os.environ)
    env["CCP_GS_RCCL_LIB"] = FAKE
    env["FAKE_RCCL_TIMEOUT_S"] = "120"
    return env


def drive(env, cases, timeout=900):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_threads_driver.py"), json.dumps(cases)],
                         capture_output=True, text=True, timeout=timeout, env=env)
    assert out.returncode == 0, out.stderr[-4000:]
    res = [json.loads(line) for line in out.stdout.splitlines() if line.startswith("{")]
    assert len(res) == len(cases), out.stderr[-4000:]
    return res


SWEEPS = [
    # world, W, H, C, ghost, iters, calls
    (2, 1000, 300, 1, 8, 13, None),
    (3, 16384, 1536, 1, 32, 40, None),          # wide enough for the in-launch edge signal (edge chunks of their own)
    (3, 777, 411, 2, 16, 29, [5, 17, 7]),       # ragged block heights, intervals straddling calls, two channels
    (4, 2048, 1024, 1, 16, 37, [20, 17]),
]
# Dirichlet-mask grids (the region of BASELINE configs[4]) in row blocks: world, W, H, C, ghost, iters, calls
MASK_SWEEPS = [
    (2, 1200, 700, 1, 12, 19, None),
    (3, 2048, 1100, 2, 16, 27, [11, 16]),
    (4, 1500, 1300, 1, 32, 40, None),
]


def test_owned_rows_equal_one_block_with_overlap_on_and_off(fake_env):
    cases = []
    for world, W, H, C, ghost, iters, calls in SWEEPS:
        for overlap in (True, False):
            c = {"kind": "sweep", "world": world, "W": W, "H": H, "C": C, "ghost": ghost, "iters": iters, "overlap": overlap}
            if calls:
                c["calls"] = calls
            cases.append(c)
    for world, W, H, C, ghost, iters, calls in MASK_SWEEPS:
        for overlap in (True, False):
            c = {"kind": "sweep", "mask": True, "world": world, "W": W, "H": H, "C": C, "ghost": ghost, "iters": iters, "overlap": overlap}
            if calls:
                c["calls"] = calls
            cases.append(c)
    for r in drive(fake_env, cases):
        c = r["case"]
        assert r["ok"], r
        assert r["bit_identical"], (c, r["rel_l2"])
        assert r["residual_close"] and r["residual_same_on_all_ranks"], c
        # the messages really went through the transport: an exchange is 2 sends + 2 receives per interior
        # neighbour pair and channel, ghost rows of 2 * pitch doubles each
        pairs = c["world"] - 1
        n_exch = set(r["exchanges"])
        assert len(n_exch) == 1, r["exchanges"]
        n = n_exch.pop()
        assert n >= 1 + c["iters"] // (c["ghost"] // 2) - 1
        assert r["sends"] == r["recvs"] == 2 * pairs * c.get("C", 1) * n, (c, r)
        pitch = ((c["W"] + 1) // 2 + 15) // 16 * 16
        assert r["bytes"] == r["recvs"] * c["ghost"] * 2 * pitch * 8, (c, r)


def test_all_reduced_stop_rule_stops_at_the_references_sweep(fake_env):
    cases = [{"kind": "stop_rule", "world": w, "W": 96, "H": 80, "ghost": g, "eps": 0.5} for w, g in ((2, 8), (3, 4), (4, 2))]
    cases += [{"kind": "stop_rule", "mask": True, "discs": 12, "world": w, "W": 160, "H": 120, "ghost": g, "eps": 5.0} for w, g in ((2, 8), (3, 2))]
    # deep checked passes (ghost 16-32: depth 8), three channels that stop at different sweeps inside different passes
    cases += [{"kind": "stop_rule", "world": w, "W": 200, "H": 144, "ghost": g, "eps": 0.5, "scale": [1e-3, 3e-4, 2e-5]} for w, g in ((2, 32), (3, 16), (4, 20))]
    cases += [{"kind": "stop_rule", "mask": True, "discs": 12, "world": 2, "W": 200, "H": 160, "ghost": 24, "eps": 5.0, "scale": [1e-3, 1e-4]}]
    for r in drive(fake_env, cases):
        assert r["ok"], r
        want = r["iterations_one_block"]
        assert want == r["iterations_oracle"], r              # the one-block solve stops where the CPU oracle stops
        for its, conv in zip(r["iterations_ranks"], r["converged"]):
            assert its == want and all(c == 1 for c in conv), r
        assert r["last_channel_bit_identical"], r["case"]
        assert r["all_channels_bit_identical"], r["case"]     # (a channel freezes at its own stop sweep, as on one block)
        for steps in r["step_ranks"]:                         # the all-reduced step: equal on every rank, equal up to summation order
            assert steps == r["step_ranks"][0]
            assert np.allclose(steps, r["step_one_block"], rtol=1e-10, atol=0.0)


def test_non_contiguous_partitions_are_rejected_on_every_rank(fake_env):
    H = 300
    cases = [
        {"kind": "bad_partition", "world": 3, "W": 256, "H": H, "ghost": 8, "parts": [[0, 100], [101, 99], [200, 100]]},   # a gap
        {"kind": "bad_partition", "world": 3, "W": 256, "H": H, "ghost": 8, "parts": [[0, 100], [200, 100], [100, 100]]},  # not in rank order
        {"kind": "bad_partition", "world": 3, "W": 256, "H": H, "ghost": 8, "parts": [[0, 100], [100, 100], [200, 90]]},   # stops short of the image
        {"kind": "bad_partition", "world": 3, "W": 256, "H": H, "ghost": 8, "parts": [[0, 100], [100, 100], [200, 100]]},  # the control: accepted
    ]
    res = drive(fake_env, cases)
    for r in res[:3]:
        assert r["ok"] and r["status"] == [1, 1, 1], r        # CCP_ERR_BAD_ARG, all ranks alike (the check is all-gathered)
    assert res[3]["ok"] and res[3]["status"] == [0, 0, 0], res[3]


def test_a_polling_wait_that_gives_up_is_an_error_not_a_wrong_ghost_row(fake_env):
    """The forced late flag: CCP_ERR_STATE (5) on the ranks whose wait gave up, right rows on the others."""
    res = drive(fake_env, [{"kind": "late_flag", "world": 2, "W": 16384, "H": 1536, "ghost": 32, "iters": 16}])[0]
    assert res["ok"], res
    for r in res["ranks"]:
        assert r["wait_mode"] == 1
        assert r["status"] == 5 or (r["status"] == 0 and r["rows_right"]), res
    assert any(r["status"] == 5 for r in res["ranks"]), res       # the forced case did fire somewhere


@pytest.mark.parametrize("world", [2, 3, 4])
def test_cpp_host_runs_every_rank_as_a_thread(fake_env, tmp_path, world):
    """tests/cpp/rowblock_driver.cpp sees only include/ccp_gs.h; rank = -1 runs all ranks as threads."""
    from coursecomputationalphotography_amd import capi
    W, H, ghost, iters = 1536, 900, 16, 45
    out = subprocess.run([os.path.join(CPP, "rowblock_driver"), str(world), "-1", str(tmp_path / "id.bin"), str(W), str(H), str(ghost), str(iters)],
                         capture_output=True, text=True, timeout=600, env=fake_env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln.split() for ln in out.stdout.splitlines() if ln.startswith("rank ")]
    assert len(lines) == world
    vals = [{t[i]: t[i + 1] for i in range(0, len(t) - 1, 2)} for t in lines]
    g = capi.Grid(W, H, 1)
    g.randomize_x(1234, 0.0, 255.0)
    g.b_from_x()
    g.fill_x(1.0)
    g.sweep(iters)
    rr, bb = g.residual_norm2()
    want_abs = g.abs_sum()[0]
    g.close()
    assert all(int(v["iterations"]) == iters for v in vals)
    assert all(int(v["exchanges"]) >= iters // (ghost // 2) for v in vals)
    assert np.isclose(sum(float(v["abs"]) for v in vals), want_abs, rtol=1e-12, atol=0.0)
    for v in vals:
        assert np.isclose(float(v["rr"]), rr[0], rtol=1e-11, atol=0.0) and np.isclose(float(v["bb"]), bb[0], rtol=1e-12, atol=0.0)


# ---- row blocks of a general CSR matrix: ccp_csr_upload_rows (BASELINE configs[4] on several GPUs) ----------
CSR_ROWS = [
    # the region matrix of configs[4] in small, unknowns in raster order: blocks talk to their neighbours only
    {"matrix": "mask", "W": 320, "H": 240, "world": 2, "iters": 12, "cg": [1e-30, 25]},
    {"matrix": "mask", "W": 400, "H": 300, "world": 3, "iters": 9, "x0": True, "slack": 3, "cuts": [0, 0.21, 0.77, 1.0], "cg": [1e-3, 4000]},
    {"matrix": "mask", "W": 512, "H": 384, "world": 4, "iters": 7, "cuts": [0, 0.5, 0.5, 0.8, 1.0], "cg": [1e-30, 30]},       # an EMPTY block
    # random symmetric pattern, the library's greedy colouring (several colours), every block coupled to every other
    {"matrix": "random", "n": 5000, "deg": 3, "world": 3, "iters": 6, "x0": True},
    {"matrix": "random", "n": 3001, "deg": 2, "world": 4, "iters": 5, "empty_rows": True, "slack": 1},
]


def test_csr_row_blocks_sweep_the_one_gpu_iterates(fake_env):
    res = drive(fake_env, [dict(c, kind="csr_rows", overlap=ov) for c in CSR_ROWS for ov in (True, False)])
    for r in res:
        c = r["case"]
        assert r["ok"], r
        assert r["bit_identical_to_one_gpu"] and r["bit_identical_to_oracle"], c
        assert r["spmv_bit_identical"], c
        assert r["residual_close"] and r["residual_same_on_all_ranks"], c
        assert r["iterations_ranks"] == [c["iters"]] * c["world"], r
        assert r["own_colours_ok"], c
        assert all(p == "sliced ELL" for p in r["path"]), r["path"]
        if "cg" in c:
            # conjugateGradient on the blocks: the dot products are added up over the ranks in another order than on one
            # GPU, so the iterates agree to rounding; a solve that stops stops at the same iteration on every rank
            assert r["cg_rel_diff"] < 1e-9, r
            assert len(set(r["cg_iterations_ranks"])) == 1 and len(set(r["cg_rnorm_ranks"])) == 1, r
            assert r["cg_converged_ranks"] == [r["cg_converged_one_gpu"]] * c["world"], r
            assert abs(r["cg_iterations_ranks"][0] - r["cg_iterations_one_gpu"]) <= (1 if r["cg_converged_one_gpu"] else 0), r
            # ... and conjugateGradientEigen (Jacobi-preconditioned) likewise
            assert r["pcg_rel_diff"] < 1e-9, r
            assert len(set(r["pcg_iterations_ranks"])) == 1 and r["pcg_converged_ranks"] == [r["pcg_converged_one_gpu"]] * c["world"], r
            assert abs(r["pcg_iterations_ranks"][0] - r["pcg_iterations_one_gpu"]) <= (1 if r["pcg_converged_one_gpu"] else 0), r
        # the reference-order sweep and insert do not shard: CCP_ERR_UNSUPPORTED (6) on every rank
        assert all(st == 6 for rank in r["unsupported"] for _, st in rank), r["unsupported"]
        # the messages went through the transport: every value sent was received, 8 bytes each, plus the halo index
        # lists and colours of the set-up (4 bytes each way per ghost)
        assert r["sends"] == r["recvs"] > 0, r
        sent = sum(r["values_sent"])
        assert r["bytes"] == 8 * sent + 2 * 4 * sum(r["ghosts"]), r
        if c["matrix"] == "mask":
            assert max(r["peers"]) <= 2, r                    # raster order: a block touches the block before and after it
            # ... through a few slices at its two ends: those are swept first and their values travel beside the rest
            assert (max(r["edge_slices"]) > 0) == c["overlap"], r
        else:
            assert max(r["peers"]) == c["world"] - 1, r
            assert max(r["edge_slices"]) == 0, r              # every slice holds a referenced row: nothing to hide behind


def test_csr_row_blocks_stop_at_the_references_sweep(fake_env):
    cases = [{"kind": "csr_rows", "matrix": "mask", "W": 200, "H": 150, "rmax": 700.0, "world": w, "iters": 500, "eps": 2.0, "overlap": w == 2}
             for w in (2, 3)]
    cases.append({"kind": "csr_rows", "matrix": "random", "n": 2500, "deg": 3, "world": 4, "iters": 500, "eps": 1e-6})
    for r in drive(fake_env, cases):
        assert r["ok"], r
        want = r["iterations_one_gpu"]
        assert 1 < want < 500 and r["converged_one_gpu"] == 1, r
        assert want == r["iterations_oracle"], r
        assert r["iterations_ranks"] == [want] * r["case"]["world"] and all(v == 1 for v in r["converged_ranks"]), r
        assert r["bit_identical_to_one_gpu"] and r["bit_identical_to_oracle"], r["case"]
        assert all(s == r["step_ranks"][0] for s in r["step_ranks"])
        assert np.isclose(r["step_ranks"][0], r["step_one_gpu"], rtol=1e-10, atol=0.0)


def test_csr_row_blocks_refuse_together(fake_env):
    base = {"kind": "csr_rows_refused", "matrix": "mask", "W": 160, "H": 120, "world": 3}
    res = drive(fake_env, [dict(base, fault="colouring", rank=1), dict(base, fault="gap", rank=2), dict(base, fault="column", rank=0),
                           dict(base, fault="none", rank=0)])
    for r in res[:3]:
        assert r["ok"], r
        assert all(st != 0 for st in r["status"]), r          # nobody hangs, nobody is left believing it holds a block
        assert all(st == 5 for st in r["solve_after"]), r     # CCP_ERR_STATE: the handle holds no matrix
    assert res[1]["status"] == [1, 1, 1], res[1]              # the partition check is all-gathered: CCP_ERR_BAD_ARG everywhere
    assert res[2]["status"][0] == 1, res[2]                   # the rank with the bad column reports its own error
    assert res[3]["ok"] and res[3]["status"] == [0, 0, 0] and res[3]["solve_after"] == [None] * 3, res[3]


def test_grid_conjugate_gradient_on_row_blocks(fake_env):
    """ccp_grid_conjugate_gradient_rowblocked — the solver the blend call sites use (PhotoMontage.cpp:613) — on 2-4 blocks,
    plain and Dirichlet-mask grids: the one-block iterates to rounding, the same stop iteration everywhere, and a sweep
    after the solve refreshes its ghost rows by itself."""
    cases = [
        {"kind": "grid_cg", "world": 2, "W": 300, "H": 200, "C": 1, "ghost": 8, "eps": 1e-30, "iters": 30},
        {"kind": "grid_cg", "world": 3, "W": 513, "H": 300, "C": 2, "ghost": 2, "eps": 1e-30, "iters": 25},
        {"kind": "grid_cg", "world": 4, "W": 640, "H": 480, "C": 1, "ghost": 4, "eps": 1e-3, "iters": 5000, "mask": True, "discs": 30},
        {"kind": "grid_cg", "world": 2, "W": 400, "H": 300, "C": 1, "ghost": 16, "eps": 1e-30, "iters": 40, "mask": True},
    ]
    for r in drive(fake_env, cases):
        c = r["case"]
        assert r["ok"], r
        assert r["rel_diff"] < 1e-9, r
        for ranks_it, ranks_conv in zip(r["iterations_ranks"], r["converged_ranks"]):
            assert ranks_it == r["iterations_ranks"][0] and ranks_conv == r["converged_ranks"][0], r
        assert r["converged_ranks"][0] == r["converged_one_block"], r
        assert all(abs(a - b) <= (1 if cv else 0) for a, b, cv in zip(r["iterations_ranks"][0], r["iterations_one_block"], r["converged_one_block"])), r
        assert r["sweep_after_solve_bit_identical"], c


def test_csr_row_blocks_random_partitions(fake_env):
    """Random sparse matrices, random cuts (empty blocks, one-row blocks, everything on one rank), worlds 2-4: the blocks'
    iterates, products and norms are the one-GPU handle's."""
    rng = np.random.Generator(np.random.MT19937(20261004))
    cases = []
    for k in range(14):
        world = int(rng.integers(2, 5))
        n = int(rng.integers(40, 2500))
        inner = sorted(int(v) for v in rng.integers(0, n + 1, world - 1))
        if k == 3:
            inner = [0] * (world - 1)                        # every row on the last rank
        if k == 7:
            inner = [1] + [n] * (world - 2) if world > 2 else [1]
        cases.append({"kind": "csr_rows", "matrix": "random", "n": n, "deg": int(rng.integers(1, 5)), "seed": int(rng.integers(1, 10**6)),
                      "world": world, "iters": int(rng.integers(1, 6)), "x0": bool(rng.integers(0, 2)), "slack": int(rng.integers(0, 3)),
                      "cuts": [0] + inner + [n], "overlap": bool(rng.integers(0, 2)), "empty_rows": bool(rng.integers(0, 2))})
    for r in drive(fake_env, cases):
        c = r["case"]
        assert r["ok"], r
        assert r["bit_identical_to_one_gpu"] and r["bit_identical_to_oracle"] and r["spmv_bit_identical"], c
        assert r["residual_close"] and r["residual_same_on_all_ranks"], c
        assert r["iterations_ranks"] == [c["iters"]] * c["world"] and r["own_colours_ok"], r


def test_row_blocked_stop_rule_random_shapes(fake_env):
    """The checked blocked passes of the row-blocked solve on random shapes, ghost depths, channel counts and scales: every
    channel stops where the one-block solve (and the oracle) stops and holds its iterate."""
    rng = np.random.Generator(np.random.MT19937(777))
    cases = []
    for k in range(14):
        world = int(rng.integers(2, 5))
        ghost = int(rng.choice([2, 4, 6, 8, 12, 16, 24, 32]))
        H = int(rng.integers(world * max(ghost, 8), world * max(ghost, 8) + 120))
        W = int(rng.integers(40, 300))
        nch = int(rng.integers(1, 4))
        scale = [float(10.0 ** rng.uniform(-5.0, -2.7)) for _ in range(nch)]
        cases.append({"kind": "stop_rule", "world": world, "W": W, "H": H, "ghost": ghost, "eps": float(rng.choice([0.5, 0.05, 2.0])),
                      "scale": scale, "mask": bool(k % 3 == 2), "discs": 10})
    for r in drive(fake_env, cases):
        c = r["case"]
        assert r["ok"], r
        want = r["iterations_one_block"]
        if not c["mask"]:
            assert want == r["iterations_oracle"], r
        for its, conv in zip(r["iterations_ranks"], r["converged"]):
            assert its == want, r
        assert r["all_channels_bit_identical"], c
