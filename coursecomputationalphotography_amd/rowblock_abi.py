"""Row-blocked multi-GPU Gauss-Seidel driven entirely through the C ABI: halo exchange, sweeps, the
all-reduced stop rule and the residual all run inside libccp_gs.so over its own RCCL communicator
(ccp_comm_*, ccp_grid_*_rowblocked; include/ccp_gs.h).  This class only hands the communicator's unique
id from rank 0 to the other ranks (torch.distributed is the host channel for that, nothing else) and
forwards calls — it is what a C++ host does in tests/cpp/rowblock_driver.cpp.
"""
from __future__ import annotations

import numpy as np


class AbiRowBlockSolver:
    def __init__(self, block, rank: int, world: int, ghost: int, dist, parts, height: int, overlap: bool = True,
                 unique_id: bytes | None = None, comm=None):
        """comm: a communicator that already exists (it is then NOT closed with the solver — one communicator
        serves several blocks in turn, e.g. the candidates of choose_ghost)."""
        from . import capi
        if ghost < 2 or ghost % 2:
            raise ValueError("ghost must be an even number >= 2 (two rows per iteration)")
        self.block, self.rank, self.world, self.ghost, self.dist = block, rank, world, ghost, dist
        self.iters_per_exchange = ghost // 2
        self.overlap = bool(overlap)
        self.owns_comm = comm is None
        if comm is None:
            if unique_id is None:
                box = [capi.comm_unique_id() if rank == 0 else None]
                if world > 1:
                    dist.broadcast_object_list(box, src=0)
                unique_id = box[0]
            comm = capi.Comm(unique_id, rank, world, block.grid.desc.device)
        self.comm = comm
        block.grid.attach_comm(self.comm)
        block.grid.set_overlap(self.overlap)

    def describe(self) -> str:
        n, mode, up, down = self.block.grid.comm_stats()
        wait = {0: "hipStreamWaitValue64 on the edge flag", 1: "a polling kernel on the edge flag", -1: "no neighbours"}[mode]
        return (f"{self.world} row blocks, ghost {self.ghost}, halo exchange every {self.iters_per_exchange} iterations by "
                f"ncclSend/ncclRecv inside libccp_gs.so (RCCL {self.comm.info()['rccl_version']})"
                + (f", beside the rest of the last pass of each interval ({wait})" if self.overlap else ", after the pass"))

    def exchange_halos(self) -> None:
        self.block.grid.exchange_halos()

    def sweep(self, iterations: int) -> None:
        self.block.grid.sweep_rowblocked(iterations)

    def gauss_seidel(self, epsilon: float = 1e-6, max_iteration: int = 1000, check_every: int = 1):
        reps = self.block.grid.gauss_seidel_rowblocked(epsilon, max_iteration, check_every)
        return max(r.iterations for r in reps), max(r.last_l1_step for r in reps)

    def rel_residual(self) -> np.ndarray:
        rr, bb = self.block.grid.residual_norm2_global()
        return np.sqrt(rr / bb)

    def close(self) -> None:
        if self.comm is not None:
            self.block.grid.synchronize()
            self.block.grid.attach_comm(None)
            if self.owns_comm:
                self.comm.close()
            self.comm = None


def _all_ok(dist, ok: bool, group=None) -> bool:
    """True iff EVERY rank reports ok: one all-reduce every rank takes part in, whatever it found."""
    import torch
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    flag = torch.tensor([0 if ok else 1], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    return int(flag.item()) == 0


def setup_abi_solver(block, rank: int, world: int, ghost: int, dist, parts, height: int, overlap: bool = True,
                     group=None, capi_module=None, comm=None):
    """The library's own communicator on every rank — or on none.  Returns (solver, None) or (None, reason).

    Every rank walks through the SAME sequence of torch.distributed collectives whatever fails where, so a rank
    that cannot load RCCL (or whose id creation fails) never leaves the others inside a mismatched collective:
      1. rank 0 creates the id — or a `None` sentinel — and the broadcast ALWAYS runs;
      2. every rank probes (ccp_comm_probe: RCCL loadable, device selectable; not collective) and the ranks agree;
      3. only then the collective ccp_comm_create (ncclCommInitRank) + ccp_grid_attach_comm, and the ranks agree again.
    `capi_module`: test seam (a stand-in for coursecomputationalphotography_amd.capi).
    `comm`: a communicator of an earlier call (solver.comm, kept alive by the caller): steps 1-2 are skipped, the
    block is attached to it (collective) and the ranks agree as in step 3."""
    capi = capi_module
    if capi is None:
        from . import capi
    if comm is not None:
        solver, err = None, None
        try:
            solver = AbiRowBlockSolver(block, rank, world, ghost, dist, parts, height, overlap=overlap, comm=comm)
        except Exception as e:
            err = f"{type(e).__name__}: {e}"
        if world > 1 and not _all_ok(dist, err is None, group):
            if solver is not None:
                solver.close()
            return None, f"ccp_grid_attach_comm failed on at least one rank ({err or 'another rank'})"
        return (solver, None) if err is None else (None, err)
    err = None
    uid = None
    if rank == 0:
        try:
            uid = capi.comm_unique_id()
        except Exception as e:                                    # CcpError (CCP_ERR_RCCL ...), OSError
            err = f"{type(e).__name__}: {e}"
    box = [uid]
    if world > 1:
        dist.broadcast_object_list(box, src=0, group=group)      # always: the id or the sentinel
    uid = box[0]
    if uid is None and err is None:
        err = "rank 0 could not create the communicator id"
    if err is None:
        try:
            capi.comm_probe(block.grid.desc.device)
        except Exception as e:
            err = f"{type(e).__name__}: {e}"
    if world > 1 and not _all_ok(dist, err is None, group):
        return None, f"ccp_comm_* unavailable on at least one rank ({err or 'another rank'})"
    if err is not None:
        return None, err
    solver = None
    try:
        solver = AbiRowBlockSolver(block, rank, world, ghost, dist, parts, height, overlap=overlap, unique_id=uid)
    except Exception as e:
        err = f"{type(e).__name__}: {e}"
    if world > 1 and not _all_ok(dist, err is None, group):
        if solver is not None:
            solver.close()
        return None, f"ccp_comm_create / ccp_grid_attach_comm failed on at least one rank ({err or 'another rank'})"
    if err is not None:
        return None, err
    return solver, None


GHOST_CANDIDATES = (32, 64, 128)


def choose_ghost(make_block, make_solver, dist, world: int, candidates=GHOST_CANDIDATES, intervals: int = 3,
                 iterations: int = 128, clock=None, group=None):
    """Pick the ghost depth (rows per side, exchanged every ghost/2 iterations) that sweeps fastest HERE: deeper
    ghosts mean fewer, larger messages and more redundant rows per interval, so the best depth depends on what an
    exchange costs on this node's links (SURVEY section 8e / hard part H4) — measured, not guessed.

    For every candidate: `make_block(ghost)` -> block (system already set up, halos not yet exchanged),
    `make_solver(block, ghost)` -> solver or None; one exchange, one warm-up interval, then the time of
    `intervals` runs of `iterations` sweeps (whole exchange intervals of every candidate) between barriers, MAX
    over the ranks, so every rank takes the same decision.  Returns (ghost, table) where table[ghost] =
    {"ms_per_iteration", "exchange_ms"}; candidates a block is too thin for are skipped on every rank alike."""
    import time
    import torch
    clock = clock or time.perf_counter
    dev = "cuda" if (world > 1 and dist.get_backend(group) == "nccl") else "cpu"

    def agree_max(v: float) -> float:
        if world == 1:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        return float(t.item())

    def barrier():
        if world > 1:
            dist.barrier(group=group)

    table = {}
    for ghost in candidates:
        block, solver, bad = None, None, 0.0
        try:
            block = make_block(ghost)
            solver = make_solver(block, ghost)
            if solver is None:
                bad = 1.0
        except Exception:
            bad = 1.0
        if agree_max(bad) > 0.0:
            if solver is not None:
                solver.close()
            if block is not None:
                block.close()
            continue
        grid = block.grid
        solver.exchange_halos()
        solver.sweep(solver.iters_per_exchange)
        grid.synchronize()
        barrier()
        t0 = clock()
        for _ in range(8):
            solver.exchange_halos()
        grid.synchronize()
        exch = agree_max((clock() - t0) / 8.0)
        # leave the block as the sweeps expect it: a fresh interval
        solver.sweep(solver.iters_per_exchange)
        grid.synchronize()
        barrier()
        t0 = clock()
        for _ in range(intervals):
            solver.sweep(iterations)
        grid.synchronize()
        secs = agree_max(clock() - t0)
        table[ghost] = {"ms_per_iteration": secs * 1e3 / (intervals * iterations), "exchange_ms": exch * 1e3}
        solver.close()
        block.close()
    if not table:
        raise ValueError("no ghost depth among the candidates fits the row blocks")
    best = min(table, key=lambda g: table[g]["ms_per_iteration"])
    return best, table
