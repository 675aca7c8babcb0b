#!/usr/bin/env python3
"""bench.py — Gauss-Seidel pixel-updates/s on the 16384x16384 5-point Poisson grid (MI355X).

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W` prints ONE
JSON line on rank 0.  A *step* is one call of the hot path — `--iters-per-step` red-black
Gauss-Seidel iterations (fixed count, epsilon = 0, no host sync inside) over the resident
system; value = pixel updates of the whole job / wall time of exactly K steps, bracketed by a
barrier + torch.cuda.synchronize() on both sides, MAX over ranks.

Workload (BASELINE.json configs[2] at N=1, configs[3] at N>1): W x H single-channel system of
SolveChannel (closed form), b = A x_true with x_true ~ U[0,255) generated ON DEVICE (resident
before the timed region), x0 = 1.0 as the reference.  N>1 row-blocks the SAME grid across the
ranks (strong scaling) with ghost-row exchange over RCCL every ghost/2 iterations.

Extra objects in the JSON line:
  roofline     — dominant kernel (k_fused_sweep: T red-black iterations per pass over the grid):
                 algorithmic bytes per launch = 24 B per pixel update x the
                 W*H*T updates one launch performs, / the average launch duration measured with
                 HIP events on the launch stream inside the timed region; peak 8 TB/s.  Because
                 the kernel keeps rows in registers across T iterations its HBM traffic
                 (`traffic`, from the committed rocprofv3 PMC pass) is far BELOW the algorithmic
                 bytes, so `frac` exceeds 1: the path runs above the 24 B/update HBM roofline.
                 `traffic_gbs` / `traffic_frac_of_peak` price the measured bytes instead: how
                 close the pass runs to what the memory system can move.
  cpu_baseline — the compiled reference header (oracle/_ref, kind "reference") or the C oracle
                 (kind "port") timed on ONE host core on a bounded sample (N=1, rank 0 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
BYTES_PER_UPDATE = 24.0        # b read 8 + neighbour-plane read 8 + x write 8


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=16384)
    ap.add_argument("--height", type=int, default=16384)
    ap.add_argument("--channels", type=int, default=1)
    ap.add_argument("--iters-per-step", type=int, default=32)
    ap.add_argument("--ghost", type=int, default=64, help="ghost rows per side (N>1); exchange every ghost/2 iterations")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default); gloo stages halos through the host (several ranks on one GPU, testing only)")
    ap.add_argument("--overlap", action="store_true", help="N>1: finish the edge rows first and exchange halos beside the rest of the pass (measured slower, see DESIGN.md)")
    ap.add_argument("--same-device", action="store_true", help="all ranks use cuda:0 (testing with --backend gloo)")
    ap.add_argument("--no-tune", action="store_true", help="skip ccp_grid_tune (use the built-in defaults)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-converge", action="store_true", help="skip the untimed iterations-to-1e-5 run")
    ap.add_argument("--no-reference-order", action="store_true", help="skip the untimed lexicographic-order run (N=1)")
    ap.add_argument("--reference-order-iters", type=int, default=128)
    ap.add_argument("--converge-cap", type=int, default=6000)
    ap.add_argument("--cpu-sample", type=int, default=4096, help="edge of the CPU baseline sample grid")
    ap.add_argument("--cpu-iters", type=int, default=96, help="CPU baseline sweeps (about 15 s of host time at the default sample)")
    return ap.parse_args()


def cpu_baseline(sample: int, iters: int):
    """Reference gaussSeidel (lexicographic, single thread by construction) on a sample grid."""
    import numpy as np
    from coursecomputationalphotography_amd import synth
    import oracle
    v, c, r = synth.poisson_csr(sample, sample)
    b, _ = synth.poisson_system(sample, sample, 1234)
    kind = "reference"
    try:
        ref = oracle.Ref()
        secs = ref.gs_csr_timed(v, c, r, b, iters)
    except (FileNotFoundError, OSError):
        kind = "port"
        m = oracle.Oracle().from_csr(v, c, r)
        t0 = time.perf_counter()
        m.gauss_seidel(b, 0.0, iters)
        secs = time.perf_counter() - t0
    ups = sample * sample * iters / secs
    return {"value": ups, "unit": "pixel-updates/s", "cores": 1, "kind": kind,
            "host_cores_available": os.cpu_count(),
            "sample": f"{sample}x{sample} single-channel Poisson, {iters} lexicographic iterations, "
                      f"{secs:.2f} s inside gaussSeidel (sweep is serial by construction)"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the Gauss-Seidel path has no CPU fallback")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    from coursecomputationalphotography_amd import rowblock

    W, H, C = args.width, args.height, args.channels
    parts = rowblock.partition_rows(H, world)
    row_begin, row_count = parts[rank]
    ghost = args.ghost if world > 1 else 0
    blk = rowblock.GridBlock(W, H, C, row_begin, row_count, ghost, local_rank)
    solver = rowblock.RowBlockSolver(blk, rank, world, max(ghost, 2), dist, overlap=args.overlap).set_partition(parts, H)
    g = blk.grid

    # synthetic system, generated on device: x_true -> b = A x_true -> x0 = 1.0
    g.randomize_x(1234, 0.0, 255.0)
    g.b_from_x()
    g.fill_x(1.0)
    if world > 1:
        solver.exchange_halos()
    g.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ips = args.iters_per_step
    tuned = None
    if not args.no_tune:
        # untimed: choose fused depth / chunk rows for this shape (speed only, results identical)
        tuned = g.tune(min(16, max(1, (ips if world == 1 else solver.iters_per_exchange) // 2)))
        g.synchronize()
    for _ in range(args.warmup):
        solver.sweep(ips)
    barrier()
    kernel_ms, launches = 0.0, 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        solver.sweep(ips)
    barrier()
    elapsed = time.perf_counter() - t0
    # HIP-event time of the LAST sweep call (one exchange interval at N>1, one step at N=1)
    ms, n_launch = g.last_timing()
    kernel_ms, launches = ms, n_launch

    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    updates = float(W) * H * C * ips * args.steps
    value = updates / elapsed

    # roofline of the dominant kernel.  The timed sweep call issued `launches` kernel launches
    # for `iters_timed` iterations over the local rows.
    iters_timed = ips if world == 1 else solver.iters_per_exchange
    fused = os.environ.get("CCP_GS_FUSE", "1") != "0" and iters_timed >= 2
    updates_per_launch = float(W) * blk.local_rows * C * iters_timed / max(launches, 1)
    avg_launch_s = (kernel_ms * 1e-3) / max(launches, 1)
    achieved = BYTES_PER_UPDATE * updates_per_launch / avg_launch_s / 1e9 if launches else None
    traffic, traffic_src = None, None
    try:   # HBM bytes per launch from the committed PMC pass of this exact configuration
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as fh:
            tj = json.load(fh)
        key = f"{W}x{H}x{C}_n{world}_ips{ips}"
        if key in tj and fused:
            traffic, traffic_src = tj[key]["hbm_bytes_per_launch"], tj[key]["source"]
    except (OSError, ValueError, KeyError):
        pass
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                "traffic_source": traffic_src,
                # what the memory system really moved (PMC pass) over the same launch duration
                "traffic_gbs": (traffic / avg_launch_s / 1e9) if (traffic and launches) else None,
                "traffic_frac_of_peak": (traffic / avg_launch_s / 1e9 / HBM_PEAK_GBS) if (traffic and launches) else None,
                "kernel": "k_fused_sweep (+ k_fused_border on a second stream, same pass)" if fused else "k_half_sweep",
                # what streaming kernels reach on this pool (tools/hbm_calib.hip, profiles/r01_hbm_calib.txt): the
                # practical ceiling for a pass that reads twice what it writes is the last figure, not 8 TB/s
                "measured_stream_ceilings_gbs": {"read": 6400, "write": 4700, "copy": 4800, "two_reads_one_write": 5400,
                                                 "source": "profiles/r01_hbm_calib.txt"},
                "iterations_per_launch": iters_timed / max(launches, 1),
                "avg_launch_ms": avg_launch_s * 1e3,
                "algorithmic_bytes_per_launch": BYTES_PER_UPDATE * updates_per_launch}

    extra = {}
    if not args.no_converge:
        # untimed: iterations until ||b - A x||_2 / ||b||_2 <= 1e-5 (continuing from the timed state)
        done = (args.warmup + args.steps) * ips
        rel = float(solver.rel_residual().max())
        trace = [[done, rel]]
        chunk = 64
        while rel > 1e-5 and done < args.converge_cap:
            solver.sweep(chunk)
            done += chunk
            rel = float(solver.rel_residual().max())
            trace.append([done, rel])
        extra["iters_to_1e-5"] = done if rel <= 1e-5 else None
        extra["rel_residual_trace"] = trace[-4:]
        extra["rel_residual_final"] = rel
        # the reference's own stop quantity, sum|x_k - x_{k-1}| (sparse-matrix.h:376), one more sweep
        extra["l1_step_after"] = [done + 1, float(solver.sweep_l1().max())]

    if world == 1 and not args.no_reference_order:
        # untimed extra: the reference's OWN sweep order (lexicographic), bit-identical iterates, on the
        # same system from the same start vector (ccp_grid_gauss_seidel_lexicographic)
        g.fill_x(1.0)
        g.gauss_seidel_lexicographic(0.0, 4, 0)                       # allocations, code load
        g.fill_x(1.0)
        rep = g.gauss_seidel_lexicographic(0.0, args.reference_order_iters, 0)[0]
        extra["reference_order"] = {
            "what": "lexicographic Gauss-Seidel (the reference's index-order sweep, sparse-matrix.h:357-370), "
                    "hyperplane-pipelined; iterates bit-identical to the reference's",
            "iterations": rep.iterations, "seconds": rep.seconds,
            "pixel_updates_per_s": float(W) * H * C * rep.iterations / rep.seconds,
            "rel_residual_after": float(solver.rel_residual().max())}

    if rank == 0:
        out = {
            "metric": "Gauss-Seidel pixel-updates/s on WxH 5-point Poisson grid",
            "value": value, "unit": "pixel-updates/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{W}x{H} single-channel 5-point Poisson (SolveChannel closed form), "
                                   f"red-black Gauss-Seidel, fixed iteration count" if C == 1 else
                                   f"{W}x{H} {C}-channel Poisson blend, red-black Gauss-Seidel",
                       "iters_per_step": ips, "channels": C,
                       "tuned": None if tuned is None else {"fused_depth": tuned[0], "rows_per_chunk": tuned[1], "ms_per_iteration": tuned[2]},
                       "partition": "single block" if world == 1 else f"{world} row blocks, ghost {ghost}, halo exchange every {ghost // 2} iterations over " + ("RCCL" if args.backend == "nccl" else "gloo (host-staged, test only)") + (", exchange beside the last pass of each interval" if solver.overlap else "")},
            "roofline_frac_of_value": value * BYTES_PER_UPDATE / 1e9 / (HBM_PEAK_GBS * world),
            "roofline": roofline,
        }
        out.update(extra)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.cpu_iters)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
