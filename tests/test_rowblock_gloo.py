"""CPU, multi-process: the row-block partition + halo-exchange logic of RowBlockSolver under
torch.distributed/gloo with world_size 2 and 3 (one process per rank, rendezvous on 127.0.0.1).
Each rank drives a numpy stand-in block (tests/rowblock_helpers.py); the gathered owned rows
must be bit-identical to the oracle's single-domain red-black sweep."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

W, H, C, GHOST = 23, 37, 2, 4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, iters, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from coursecomputationalphotography_amd import rowblock, synth
    from rowblock_helpers import NumpyBlock
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b = np.stack([synth.poisson_system(W, H, 40 + ch)[0] for ch in range(C)])
        parts = rowblock.partition_rows(H, world)
        rb, rc = parts[rank]
        blk = NumpyBlock(W, H, C, rb, rc, GHOST, b)
        solver = rowblock.RowBlockSolver(blk, rank, world, GHOST, dist).set_partition(parts, H)
        solver.exchange_halos()
        solver.sweep(iters)
        l1 = solver.sweep_l1()                      # iteration iters+1, all-reduced step
        res = solver.rel_residual()
        it2, eps2 = solver.gauss_seidel(epsilon=float(l1.max()) * 0.9, max_iteration=50, check_every=1)
        q.put((rank, rb, blk.owned(), l1, res, it2, eps2))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_row_blocks_equal_single_domain(orc, world):
    import oracle
    from coursecomputationalphotography_amd import synth
    iters = 7
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, iters, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    v, c, r = synth.poisson_csr(W, H)
    col = oracle.grid_colour(W, H)
    it2, eps2 = results[0][5], results[0][6]
    total = iters + 1 + it2
    for ch in range(C):
        b = synth.poisson_system(W, H, 40 + ch)[0]
        want, _, _ = orc.multicolour_gauss_seidel(v, c, r, col, b, 0.0, total)
        got = np.concatenate([res[2][ch] for res in results]).ravel()
        assert np.array_equal(got, want), (world, ch)
        # all-reduced L1 step of iteration iters+1 and the relative residual after it
        _, _, eps = orc.multicolour_gauss_seidel(v, c, r, col, b, 0.0, iters + 1)
        x1, _, _ = orc.multicolour_gauss_seidel(v, c, r, col, b, 0.0, iters + 1)
        assert abs(results[0][3][ch] - eps) <= 1e-12 * eps
        m = orc.from_csr(v, c, r)
        assert abs(results[0][4][ch] - m.rel_residual(b, x1)) <= 1e-12
    for res in results:                             # every rank saw the same reduced numbers
        assert np.array_equal(res[3], results[0][3]) and res[5] == it2


def test_partition_rows():
    from coursecomputationalphotography_amd.rowblock import partition_rows
    assert partition_rows(16384, 8) == [(i * 2048, 2048) for i in range(8)]
    parts = partition_rows(37, 3)
    assert parts == [(0, 13), (13, 12), (25, 12)]
    assert sum(c for _, c in parts) == 37


# ---- the ranks agree on the library's own communicator, or fall back together ---------------------------------
class _FakeGrid:
    class desc:
        device = 0


class _FakeBlock:
    grid = _FakeGrid()


def _agree_worker(rank, world, port, scenario, q):
    """setup_abi_solver under gloo with a stand-in capi: whatever fails on whichever rank, every rank must come
    back (no mismatched collective), all with the same decision."""
    import sys
    import types
    sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from coursecomputationalphotography_amd import rowblock_abi
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if scenario == "no_rccl_library":
            # the real binding with an RCCL that cannot be loaded: ccp_comm_unique_id / ccp_comm_probe report
            # CCP_ERR_RCCL (no GPU is needed to get that far)
            os.environ["CCP_GS_RCCL_LIB"] = "/nonexistent/librccl.so.1"
            capi = None
        else:
            capi = types.SimpleNamespace()

            def unique_id():
                if scenario == "id_fails_on_rank0":
                    raise RuntimeError("ncclGetUniqueId failed")
                return b"x" * 128

            def probe(device):
                if scenario == "probe_fails_on_rank1" and rank == 1:
                    raise RuntimeError("librccl.so.1 not loadable here")

            capi.comm_unique_id, capi.comm_probe = unique_id, probe
            if scenario == "create_fails_on_rank1":
                # AbiRowBlockSolver.__init__ -> capi.Comm(...): succeed on rank 0 (a dummy), fail on rank 1
                class Comm:
                    def __init__(self, *a):
                        if rank == 1:
                            raise RuntimeError("ncclCommInitRank failed")

                    def close(self):
                        pass
                capi.Comm = Comm
                _FakeGrid.attach_comm = lambda self, c: None
                _FakeGrid.set_overlap = lambda self, on: None
                _FakeGrid.synchronize = lambda self: None
                import coursecomputationalphotography_amd as pkg
                pkg.capi = capi                                   # `from . import capi` inside AbiRowBlockSolver
                sys.modules["coursecomputationalphotography_amd.capi"] = capi
        solver, why = rowblock_abi.setup_abi_solver(_FakeBlock(), rank, world, 4, dist, [(0, 8), (8, 8)], 16, capi_module=capi)
        q.put((rank, solver is None, why))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("scenario", ["no_rccl_library", "id_fails_on_rank0", "probe_fails_on_rank1", "create_fails_on_rank1"])
def test_ranks_fall_back_together_when_the_communicator_cannot_be_set_up(scenario):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_agree_worker, args=(r, world, port, scenario, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(none for _, none, _ in results), results          # nobody kept a half-built communicator
    assert all(why for _, _, why in results), results            # and everybody knows why (bench.py's halo_note)


# ---- ghost depth chosen from measured times (rowblock_abi.choose_ghost) -----------------------------------------
class _ClockedSolver:
    """A stand-in whose exchange and sweeps advance a virtual clock: an exchange costs `latency + rows * per_row`,
    a sweep of k iterations costs the rows it marches (owned + the ghost rows still valid, both sides)."""

    def __init__(self, clock, ghost, rows, latency, per_row_sweep):
        self.clock, self.ghost, self.rows, self.latency, self.per_row = clock, ghost, rows, latency, per_row_sweep
        self.iters_per_exchange = ghost // 2
        self.since = 0
        self.closed = False

    def exchange_halos(self):
        self.clock[0] += self.latency + self.ghost * 1e-7
        self.since = 0

    def sweep(self, iterations):
        left = iterations
        while left > 0:
            if self.since >= self.iters_per_exchange:
                self.exchange_halos()
            room = min(left, self.iters_per_exchange - self.since)
            self.clock[0] += room * (self.rows + 2 * self.ghost) * self.per_row
            self.since += room
            left -= room

    def close(self):
        self.closed = True


class _ClockedBlock:
    class grid:
        @staticmethod
        def synchronize():
            pass

    def __init__(self):
        self.closed = False

    def close(self):
        self.closed = True


@pytest.mark.parametrize("latency, want", [(0.0, 32), (2e-3, 128), (5e-5, 64)])
def test_ghost_depth_follows_the_measured_exchange_cost(latency, want):
    """Cheap exchanges -> the shallowest ghosts (least redundant rows); expensive ones -> the deepest (fewest messages);
    in between the middle.  Virtual clock: the decision depends on the timings alone."""
    from coursecomputationalphotography_amd import rowblock_abi
    clock = [0.0]
    made = []

    def make_block(ghost):
        made.append(_ClockedBlock())
        return made[-1]

    def make_solver(block, ghost):
        return _ClockedSolver(clock, ghost, rows=2048, latency=latency, per_row_sweep=1e-8)

    best, table = rowblock_abi.choose_ghost(make_block, make_solver, dist=None, world=1, clock=lambda: clock[0])
    assert best == want, table
    assert set(table) == {32, 64, 128} and all(b.closed for b in made)
    assert all(t["exchange_ms"] >= latency * 1e3 for t in table.values())


def test_ghost_candidates_a_block_is_too_thin_for_are_skipped():
    from coursecomputationalphotography_amd import rowblock_abi
    clock = [0.0]

    def make_block(ghost):
        if ghost > 64:
            raise ValueError("row block thinner than the ghost depth")
        return _ClockedBlock()
    best, table = rowblock_abi.choose_ghost(make_block, lambda b, g: _ClockedSolver(clock, g, 100, 1.0, 1e-8), dist=None, world=1,
                                            clock=lambda: clock[0])
    assert best == 64 and set(table) == {32, 64}


def _ghost_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from coursecomputationalphotography_amd import rowblock_abi
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        clock = [0.0]
        # rank 1 sees a slow link, rank 0 a free one: the MAX over ranks decides, both take the same depth;
        # rank 1 cannot build the 128-row candidate: it is skipped on BOTH ranks
        lat = 2e-3 if rank == 1 else 0.0

        def make_block(ghost):
            if rank == 1 and ghost == 128:
                raise ValueError("too thin")
            return _ClockedBlock()
        best, table = rowblock_abi.choose_ghost(make_block, lambda b, g: _ClockedSolver(clock, g, 2048, lat, 1e-8), dist, world,
                                                clock=lambda: clock[0])
        q.put((rank, best, sorted(table)))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_ranks_take_the_same_ghost_depth():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ghost_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == 64 and res[0][2] == res[1][2] == [32, 64]
