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
                 unique_id: bytes | None = None):
        from . import capi
        if ghost < 2 or ghost % 2:
            raise ValueError("ghost must be an even number >= 2 (two rows per iteration)")
        self.block, self.rank, self.world, self.ghost, self.dist = block, rank, world, ghost, dist
        self.iters_per_exchange = ghost // 2
        self.overlap = bool(overlap)
        if unique_id is None:
            box = [capi.comm_unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(box, src=0)
            unique_id = box[0]
        self.comm = capi.Comm(unique_id, rank, world, block.grid.desc.device)
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
                     group=None, capi_module=None):
    """The library's own communicator on every rank — or on none.  Returns (solver, None) or (None, reason).

    Every rank walks through the SAME sequence of torch.distributed collectives whatever fails where, so a rank
    that cannot load RCCL (or whose id creation fails) never leaves the others inside a mismatched collective:
      1. rank 0 creates the id — or a `None` sentinel — and the broadcast ALWAYS runs;
      2. every rank probes (ccp_comm_probe: RCCL loadable, device selectable; not collective) and the ranks agree;
      3. only then the collective ccp_comm_create (ncclCommInitRank) + ccp_grid_attach_comm, and the ranks agree again.
    `capi_module`: test seam (a stand-in for coursecomputationalphotography_amd.capi)."""
    capi = capi_module
    if capi is None:
        from . import capi
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
