"""Row-blocked multi-GPU Gauss-Seidel: one process per GPU, image rows split into contiguous
blocks, ghost rows exchanged with the two neighbour ranks over RCCL (torch.distributed
backend "nccl" on ROCm) or gloo (CPU tests).

Design (hard part H4): instead of one ghost row per colour half-sweep — four
latency-bound point-to-point messages per iteration — every block keeps ``ghost = 2k`` ghost
rows per side and exchanges them once every ``k`` iterations.  Between two exchanges the
kernels recompute the ghost rows redundantly, validity receding one row per half-sweep
(``ccp_grid_sweep`` shrinks its row range accordingly), which is numerically exact for
red-black ordering: the owned rows are bit-identical to the single-GPU sweep.  A message is
``2k`` whole image rows (``2k * 2*pitch`` doubles, contiguous in the row-split layout), e.g.
2 MiB at W=16384, k=8 — one send and one receive per neighbour per k iterations.

The numerical work is the C ABI's (``capi.Grid``); this module only decides who owns which
rows and moves halos.  Any object with the small ``Block`` surface below can be driven, which
is how the CPU tests exercise the exchange logic under gloo.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np


def partition_rows(height: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous row blocks [(row_begin, row_count)]; the first ``height % world`` ranks get
    one extra row."""
    base, extra = divmod(height, world)
    out, start = [], 0
    for r in range(world):
        cnt = base + (1 if r < extra else 0)
        out.append((start, cnt))
        start += cnt
    return out


class _DevArray:
    """Zero-copy view of device memory for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr: int, shape: Tuple[int, ...]):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": "<f8", "data": (ptr, False),
                                         "version": 2, "strides": None}


class GridBlock:
    """capi.Grid plus a torch view of its x rows: x_rows[ch, l] is the 2*pitch-double storage
    of local image row l."""

    def __init__(self, W, H, channels, row_begin, row_count, ghost, device_index):
        import torch
        from . import capi
        self.grid = capi.Grid(W, H, channels, row_begin, row_count, ghost, device_index)
        lay = self.grid.layout
        self.ghost_top, self.ghost_bottom = lay.ghost_top, lay.ghost_bottom
        self.local_rows, self.row_count = lay.local_rows, row_count
        self.x_rows = torch.as_tensor(_DevArray(lay.x_dev, (channels, lay.local_rows, 2 * lay.pitch)),
                                      device=torch.device("cuda", device_index))
        self._side = None

    def close(self) -> None:
        self.x_rows = None
        self.grid.close()

    def sweep(self, iterations: int) -> None:
        self.grid.sweep(iterations)

    def sweep_edges_first(self, iterations: int, edge_rows: int) -> None:
        self.grid.sweep_edges_first(iterations, edge_rows)

    def side_stream(self):
        """The torch stream halo messages are issued on; it waits for the edge rows only."""
        import torch
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.x_rows.device)
        return self._side

    def side_stream_wait_edges(self) -> None:
        self.grid.stream_wait_edges(self.side_stream().cuda_stream)

    def sweep_l1(self) -> np.ndarray:
        return self.grid.sweep_l1()

    def halo_refreshed(self) -> None:
        self.grid.halo_refreshed()

    def residual_norm2(self):
        return self.grid.residual_norm2()


class RowBlockSolver:
    """Drives one block per rank.  ``dist`` is torch.distributed (already initialised)."""

    def __init__(self, block, rank: int, world: int, ghost: int, dist, group=None, overlap: bool = False):
        if ghost < 2 or ghost % 2:
            raise ValueError("ghost must be an even number >= 2 (two rows per iteration)")
        self.block, self.rank, self.world, self.ghost, self.dist, self.group = block, rank, world, ghost, dist, group
        self.iters_per_exchange = ghost // 2
        self.since_exchange = 0
        # overlap: the pass that uses up the ghost rows finishes the rows the neighbours need first
        # and the exchange runs on a side stream beside the rest of that pass (device blocks only).
        # Off by default: on MI355X the extra band launches cost a 2048-row block 0.17-0.25 ms per
        # interval (a pass has a floor of one wave lifetime however few tiles it has), more than the
        # ~0.13 ms message they could hide (tools/rank_block_bench.py).
        self.overlap = overlap and world > 1 and hasattr(block, "sweep_edges_first")
        self._views = None
        if world > 1 and block.row_count < ghost:
            raise ValueError(f"row block of {block.row_count} rows is thinner than the ghost depth {ghost}")

    def describe(self) -> str:
        be = self.dist.get_backend(self.group) if self.world > 1 else "none"
        how = "RCCL point-to-point via torch.distributed" if be == "nccl" else f"{be} (host-staged, test only)"
        return (f"{self.world} row blocks, ghost {self.ghost}, halo exchange every {self.iters_per_exchange} iterations over {how}"
                + (", exchange beside the last pass of each interval" if self.overlap else ""))

    def close(self) -> None:
        pass

    # -- halo exchange ---------------------------------------------------------------------
    def exchange_halos(self, after_edges: bool = False) -> None:
        """Send the outermost owned rows to the neighbours' ghost rows (both directions).
        after_edges: the sweeps were issued with sweep_edges_first — run the messages on the side
        stream, which waits for the edge rows only, and let the main stream wait for the result."""
        blk, dist = self.block, self.dist
        if self.world > 1 and after_edges:
            import torch
            side = blk.side_stream()
            blk.side_stream_wait_edges()
            with torch.cuda.stream(side):
                self._move_halos()
            torch.cuda.current_stream(blk.x_rows.device).wait_stream(side)
        elif self.world > 1:
            self._move_halos()
        blk.halo_refreshed()
        self.since_exchange = 0

    def _move_halos(self) -> None:
        blk, dist = self.block, self.dist
        x = blk.x_rows
        if self._views is None:                     # the row views never change: build them once
            C = x.shape[0]
            gt, gb = blk.ghost_top, blk.ghost_bottom
            own_lo, own_hi = gt, gt + blk.row_count
            sends, recvs = [], []                   # (tensor view, peer)
            for ch in range(C):
                if self.rank > 0:      # upper neighbour: my top owned rows <-> my top ghosts
                    n_send = self._peer_ghost_bottom(self.rank - 1)
                    sends.append((x[ch, own_lo:own_lo + n_send], self.rank - 1))
                    recvs.append((x[ch, 0:gt], self.rank - 1))
                if self.rank < self.world - 1:
                    n_send = self._peer_ghost_top(self.rank + 1)
                    sends.append((x[ch, own_hi - n_send:own_hi], self.rank + 1))
                    recvs.append((x[ch, own_hi:own_hi + gb], self.rank + 1))
            self._views = (sends, recvs)
        sends, recvs = self._views
        # gloo cannot move device memory: stage through host copies (CPU tests of the GPU
        # path with several ranks on one card; RCCL sends the device rows directly)
        staged = x.is_cuda and dist.get_backend(self.group) == "gloo"
        if x.is_cuda and dist.get_backend(self.group) == "nccl":
            # Stream contract of this (torch.distributed) path: ProcessGroupNCCL orders its send/receive kernels
            # after the CURRENT torch stream, and the grid's kernels run on the handle's stream — the two must be
            # the same stream, which is only the case while the handle is on the null stream (torch's default
            # stream on ROCm).  The path that does not depend on this is halo="abi" (ccp_grid_*_rowblocked).
            if getattr(getattr(blk, "grid", None), "stream_handle", 0) != 0:
                raise RuntimeError("torch.distributed halo exchange needs the grid handle on the null stream (torch's "
                                   "default); use rowblock.make_solver(..., halo='abi') with a stream of your own")
        out = [(t.cpu() if staged else t, p) for t, p in sends]
        inn = [((t.new_empty(t.shape, device="cpu") if staged else t), p) for t, p in recvs]
        ops = [dist.P2POp(dist.isend, t, p, self.group) for t, p in out]
        ops += [dist.P2POp(dist.irecv, t, p, self.group) for t, p in inn]
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if staged:
            for (dst, _), (src, _) in zip(recvs, inn):
                dst.copy_(src)

    # ghost depths of the neighbours (they may be clipped by the image border)
    def _peer_ghost_top(self, peer: int) -> int:
        return min(self.ghost, self._parts[peer][0])

    def _peer_ghost_bottom(self, peer: int) -> int:
        begin, cnt = self._parts[peer]
        return min(self.ghost, self._height - begin - cnt)

    def set_partition(self, parts: Sequence[Tuple[int, int]], height: int) -> "RowBlockSolver":
        self._parts, self._height = list(parts), height
        return self

    # -- sweeps ----------------------------------------------------------------------------
    def sweep(self, iterations: int) -> None:
        """``iterations`` red-black sweeps with a halo exchange every ghost/2 iterations.
        Precondition: halos valid for ``iters_per_exchange - since_exchange`` more iterations."""
        left = iterations
        while left > 0:
            if self.world > 1 and self.since_exchange >= self.iters_per_exchange:
                self.exchange_halos()
            room = left if self.world == 1 else min(left, self.iters_per_exchange - self.since_exchange)
            if self.overlap and self.since_exchange + room == self.iters_per_exchange:
                # these sweeps use up the ghost rows: exchange right away, beside their last pass
                self.block.sweep_edges_first(room, self.ghost)
                self.since_exchange += room
                self.exchange_halos(after_edges=True)
            else:
                self.block.sweep(room)
                self.since_exchange += room
            left -= room

    def sweep_l1(self) -> np.ndarray:
        """One sweep returning the GLOBAL sum|x_new - x_old| per channel (all-reduced)."""
        import torch
        if self.world > 1 and self.since_exchange >= self.iters_per_exchange:
            self.exchange_halos()
        local = self.block.sweep_l1()
        self.since_exchange += 1
        if self.world == 1:
            return local
        t = torch.from_numpy(np.asarray(local, dtype=np.float64).copy())
        if self.dist.get_backend(self.group) == "nccl":
            t = t.cuda()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def gauss_seidel(self, epsilon: float = 1e-6, max_iteration: int = 1000, check_every: int = 1):
        """The reference loop (sparse-matrix.h:350-380) on the partitioned system, single RHS
        semantics applied to the channel-summed step (all channels stop together).
        Returns (iterations, last_l1_step)."""
        eps, cnt = 10.0, 0
        while eps > epsilon and cnt < max_iteration:
            if check_every > 0 and (cnt + 1) % check_every == 0:
                eps = float(np.max(self.sweep_l1()))
                cnt += 1
            else:
                run = 1 if check_every > 0 else max_iteration - cnt
                self.sweep(run)
                cnt += run
        return cnt, eps

    def rel_residual(self) -> np.ndarray:
        """sqrt(sum (b-Ax)^2 / sum b^2) per channel over the whole image."""
        import torch
        if self.world > 1 and self.since_exchange > 0:
            self.exchange_halos()
        rr, bb = self.block.residual_norm2()
        t = torch.from_numpy(np.concatenate([rr, bb]).astype(np.float64))
        if self.world > 1:
            if self.dist.get_backend(self.group) == "nccl":
                t = t.cuda()
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            t = t.cpu()
        v = t.numpy()
        C = len(rr)
        return np.sqrt(v[:C] / v[C:])


def make_solver(block, rank: int, world: int, ghost: int, dist, parts, height: int, halo: str = "torch",
                overlap: bool = False, group=None):
    """The row-block driver of one rank: halo = "abi" moves halos and norms inside libccp_gs.so over its
    own RCCL communicator (ccp_comm_*, ccp_grid_gauss_seidel_rowblocked); "torch" is the
    torch.distributed point-to-point path (gloo tests, several ranks on one card)."""
    if halo == "abi" and world > 1:
        from .rowblock_abi import AbiRowBlockSolver
        return AbiRowBlockSolver(block, rank, world, ghost, dist, parts, height, overlap=overlap)
    return RowBlockSolver(block, rank, world, ghost, dist, group=group, overlap=False).set_partition(parts, height)
