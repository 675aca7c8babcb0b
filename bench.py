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

Objects in the JSON line beyond the contract keys:
  roofline     — dominant kernel k_fused_sweep<T> (one PASS = T red-black iterations over the grid,
                 one HBM round trip).  Bytes model per launch: 24 B per pixel per pass (x read 8 +
                 b read 8 + x write 8; SURVEY §8d's 24 B/update re-based to the pass, i.e. 24/T B per
                 update) x the pixels of the local block.  achieved = that / the average launch
                 duration, measured with HIP events on the launch stream over exactly the timed
                 steps (ccp_grid_region_begin/_end); peak 8 TB/s; frac = achieved / peak <= 1.
                 `traffic` = HBM bytes per launch from the committed rocprofv3 PMC passes
                 (FETCH_SIZE doubled per the gfx950 correction, WRITE_SIZE as is) of the SAME shape and
                 tiling (profiles/r02_traffic.json, keyed WxHxC_T<depth>_R<rows>) — null when this
                 run's tiling has no committed profile.  `x_over_streaming_roofline` is the old
                 headline: value x 24 B / 8 TB/s, how far above the one-iteration-per-round-trip
                 roofline the temporal blocking runs.
  parity_check — the timed tiling re-checked in this very run: the same iteration count through the
                 independent in-place half-sweep kernels (ccp_grid_set_fused(0)) must give the same
                 |x| checksum, residual sums and row bands, bit for bit.
  iters_to_1e-5 — FIRST k with ||b - A x_k||_2 / ||b||_2 <= 1e-5 from x0 = 1 (own untimed solve,
                 independent of --steps/--warmup).
  configs      — BASELINE.json configs[0], [1], [4] measured in the same run (N=1), each with its own
                 bytes model, roofline fraction and CPU baseline.
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
BYTES_PER_PIXEL_PASS = 24.0    # the same three streams, once per PASS of the temporally blocked kernel
MIN_BYTES_PER_PIXEL_PASS = 20.0  # what an unchecked pass must move: black x 4 + b 8 + x write 8 (its first half-sweep overwrites red)
# Fixed tiling of the headline shape (depth T, rows a wave finalises per pass): what ccp_grid_tune picks on
# MI355X for this shape, pinned so that every run uses the tiling the committed PMC traffic profile and the
# full-width oracle test (tests/test_gpu_fullsize.py) were made with.  --tune re-times it on the box.
DEFAULT_TILING = {(16384, 16384, 1): (8, 364)}
CONFIG1_TILING = (8, 140)
TRAFFIC_FILES = [os.path.join(ROOT, "profiles", "r03_traffic.json"), os.path.join(ROOT, "profiles", "r02_traffic.json")]


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
    ap.add_argument("--halo", default="abi", choices=["abi", "torch"],
                    help="N>1: abi = halo exchange + norms inside libccp_gs.so over its own RCCL communicator (default); "
                         "torch = torch.distributed point-to-point (the only choice with --backend gloo)")
    ap.add_argument("--no-overlap", action="store_true", help="N>1: exchange halos after the pass instead of beside it")
    ap.add_argument("--same-device", action="store_true", help="all ranks use cuda:0 (testing with --backend gloo)")
    ap.add_argument("--tune", action="store_true", help="time depth / chunk-row candidates on this box instead of the pinned tiling")
    ap.add_argument("--depth", type=int, default=0, help="fused depth T (with --rows-per-chunk: overrides the pinned tiling)")
    ap.add_argument("--rows-per-chunk", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-converge", action="store_true", help="skip the untimed iterations-to-1e-5 solve")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run parity check against the in-place kernels")
    ap.add_argument("--no-reference-order", action="store_true", help="skip the untimed lexicographic-order run (N=1)")
    ap.add_argument("--no-configs", action="store_true", help="skip BASELINE configs[0], [1], [4] (N=1)")
    ap.add_argument("--reference-order-iters", type=int, default=128)
    ap.add_argument("--converge-cap", type=int, default=6000)
    ap.add_argument("--cpu-sample", type=int, default=4096,
                    help="edge of the CPU baseline grid.  16384 times the headline system itself (BASELINE.md section 3: 3 sweeps): "
                         "~45 GB of host memory and ~5 minutes, almost all of it building the 1.34e9-entry system on one host core, "
                         "which is why the default run times a 4096^2 sample instead")
    ap.add_argument("--cpu-iters", type=int, default=0, help="CPU baseline sweeps (default: 96 at the sample, 3 at 16384^2)")
    return ap.parse_args()


def cpu_gs_timed(v, c, r, b, iters):
    """Seconds inside the reference gaussSeidel (compiled header) or, without it, the C oracle; one core."""
    import oracle
    try:
        return "reference", oracle.Ref().gs_csr_timed(v, c, r, b, iters)
    except (FileNotFoundError, OSError):
        m = oracle.Oracle().from_csr(v, c, r)
        t0 = time.perf_counter()
        m.gauss_seidel(b, 0.0, iters)
        return "port", time.perf_counter() - t0


def mem_available_gb():
    try:
        with open("/proc/meminfo") as fh:
            for line in fh:
                if line.startswith("MemAvailable:"):
                    return int(line.split()[1]) / 1048576.0
    except OSError:
        pass
    return 0.0


def cpu_baseline(sample: int, iters: int):
    """Reference gaussSeidel (lexicographic, single thread by construction) on a sample of the workload: the same
    closed-form system (the oracle's generator, orc_poisson_csr) with b = A x_true as everywhere else.  The default
    sample is 4096^2 x 96 sweeps (~15 s inside gaussSeidel): the 16384^2 system itself (--cpu-sample 16384, 3 sweeps)
    costs more than 19 minutes of one host core before its first sweep on the GPU box (1.34e9 entries, 16 GB, built,
    multiplied and ingested serially: a run of round 3 was stopped by the 20-minute limit of a call while still
    building) against the ~1 minute of the whole run.  The sample flatters the CPU if anything: 512^2 runs at 1.36e8, 4096^2
    (1.3 GB of CSR) at 1.23-1.27e8, and 8192^2 at 2.0e7 in the build container (3 sweeps, 10.3 s inside gaussSeidel)."""
    import oracle
    from coursecomputationalphotography_amd import synth
    note = ""
    if sample > 8192 and mem_available_gb() < 80.0:
        note = f" (a {sample}^2 system needs ~45 GB of host memory, {mem_available_gb():.0f} GB available: sample instead)"
        sample = 4096
    if iters <= 0:
        iters = 3 if sample > 8192 else 96
    t0 = time.perf_counter()
    v, c, r = oracle.Oracle().poisson_csr(sample, sample)
    b = synth.poisson_apply(sample, sample, synth.x_true(sample * sample, 1234))
    setup = time.perf_counter() - t0
    kind, secs = cpu_gs_timed(v, c, r, b, iters)
    return {"value": float(sample) * sample * iters / secs, "unit": "pixel-updates/s", "cores": 1, "kind": kind,
            "host_cores_available": os.cpu_count(),
            "sample": f"{sample}x{sample} single-channel Poisson{' (the headline system itself)' if sample == 16384 else ' (a sample: building the 16384^2 system on one host core takes minutes)'}, "
                      f"{iters} lexicographic iterations, {secs:.2f} s inside gaussSeidel (sweep is serial by construction; "
                      f"{setup:.1f} s to build the system on the host){note}"}


def load_traffic(key):
    for path in TRAFFIC_FILES:
        try:
            with open(path) as fh:
                hit = json.load(fh).get(key)
        except (OSError, ValueError):
            hit = None
        if hit:
            return hit
    return None


def traffic_fields(key, seconds_per_launch):
    """`traffic` of a configs[] kernel from the committed PMC summaries (HBM bytes per launch), or nulls."""
    tr = load_traffic(key)
    if not tr:
        return {"traffic": None, "traffic_key": key}
    t = tr["hbm_bytes_per_launch"]
    return {"traffic": t, "traffic_key": key, "traffic_source": tr["source"],
            "traffic_gbs": t / seconds_per_launch / 1e9 if seconds_per_launch > 0 else None,
            "traffic_frac_of_peak": t / seconds_per_launch / 1e9 / HBM_PEAK_GBS if seconds_per_launch > 0 else None}


def roofline_of_pass(W, rows, C, T, R, ms_per_launch, value, world):
    """The roofline object of one k_fused_sweep launch over a W x rows x C block."""
    pixels = float(W) * rows * C
    model = BYTES_PER_PIXEL_PASS * pixels
    s = ms_per_launch * 1e-3
    achieved = model / s / 1e9
    key = f"{W}x{rows}x{C}_T{T}_R{R}"
    tr = load_traffic(key)
    traffic = tr["hbm_bytes_per_launch"] if tr else None
    return {"bound": "hbm", "kernel": f"k_fused_sweep<{T},0,2> (+ k_fused_border<{T},0,2> on a second stream, same pass)",
            "model": f"{BYTES_PER_PIXEL_PASS:.0f} B per pixel per pass (x read 8 + b read 8 + x write 8); a pass = {T} iterations, "
                     f"i.e. {BYTES_PER_PIXEL_PASS / T:.1f} B per pixel update",
            "bytes_per_launch": model, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "min_bytes_per_launch": MIN_BYTES_PER_PIXEL_PASS * pixels,
            "frac_of_min_bytes": MIN_BYTES_PER_PIXEL_PASS * pixels / s / 1e9 / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_key": key, "traffic_source": tr["source"] if tr else None,
            "traffic_gbs": (traffic / s / 1e9) if traffic else None,
            "traffic_frac_of_peak": (traffic / s / 1e9 / HBM_PEAK_GBS) if traffic else None,
            "avg_launch_ms": ms_per_launch, "iterations_per_launch": T,
            "x_over_streaming_roofline": value * BYTES_PER_UPDATE / 1e9 / (HBM_PEAK_GBS * world),
            # what streaming kernels reach on this pool (tools/hbm_calib.hip, profiles/r01_hbm_calib.txt)
            "measured_stream_ceilings_gbs": {"read": 6400, "write": 4700, "copy": 4800, "two_reads_one_write": 5400,
                                             "source": "profiles/r01_hbm_calib.txt"}}


def first_k_below(g, tol, cap, coarse=8):
    """FIRST k with ||b - A x_k|| / ||b|| <= tol from x0 = 1: coarse search in steps of `coarse`, then the last
    `coarse` iterations one by one from a fresh solve.  Returns (k or None, trace)."""
    import numpy as np

    def rel():
        rr, bb = g.residual_norm2()
        return float(np.sqrt(rr / bb).max())

    g.fill_x(1.0)
    k, r, trace = 0, rel(), []
    while r > tol and k < cap:
        g.sweep(coarse)
        k += coarse
        r = rel()
        trace.append([k, r])
    if r > tol:
        return None, trace[-4:]
    g.fill_x(1.0)
    lo = k - coarse
    if lo > 0:
        g.sweep(lo)
    k, r = lo, rel()
    while r > tol:
        g.sweep(1)
        k += 1
        r = rel()
    trace.append([k, r])
    return k, trace[-5:]


def parity_against_in_place(g, W, H, iters):
    """The state after `iters` fused iterations from x0 = 1 is in g; redo them with the in-place half-sweep
    kernels on the same handle and compare checksums, residual sums and row bands bit for bit."""
    import numpy as np
    bands = [0, H // 2 - 2, H - 4]
    s0 = g.abs_sum().copy()
    rr0, bb0 = g.residual_norm2()
    rows0 = [g.get_x(0, y, 4).copy() for y in bands]
    g.fill_x(1.0)
    g.set_fused(False)
    g.sweep(iters)
    g.set_fused(True)
    s1 = g.abs_sum()
    rr1, bb1 = g.residual_norm2()
    rows1 = [g.get_x(0, y, 4) for y in bands]
    return {"against": "in-place half-sweep kernels (k_half_sweep, ccp_grid_set_fused(0)), same handle, same b, x0 = 1",
            "iterations": iters,
            "abs_sum_equal": bool(np.array_equal(s0, s1)),
            "residual_sums_equal": bool(np.array_equal(rr0, rr1) and np.array_equal(bb0, bb1)),
            "bands_equal": bool(all(np.array_equal(a, b) for a, b in zip(rows0, rows1))),
            "rel_residual": float(np.sqrt(rr1 / bb1).max())}


# k_lex_wg at 8 sweeps per pass: b row of 76 columns per 62 pixels, x row of 63, x write, edge values both ways
LEX_WG_BYTES_PER_UPDATE = (76.0 / 62.0 * 8.0 + 63.0 / 62.0 * 8.0 + 8.0) / 8.0 + 2.0 * 16.0 / 62.0


def config0(capi):
    """BASELINE configs[0]: 512x512 single channel, the reference's lexicographic order (plumbing case)."""
    import numpy as np
    from coursecomputationalphotography_amd import synth
    W = H = 512
    iters = 100
    b, _ = synth.poisson_system(W, H, 1234)
    v, c, r = synth.poisson_csr(W, H)
    g = capi.Grid(W, H, 1)
    g.set_b(b)
    g.fill_x(1.0)
    g.gauss_seidel_lexicographic(0.0, 4, 0)
    g.fill_x(1.0)
    rep = g.gauss_seidel_lexicographic(0.0, iters, 0)[0]
    x = g.get_x().ravel()
    g.close()
    kind, secs = cpu_gs_timed(v, c, r, b, iters)
    import oracle
    want, _, _ = oracle.Oracle().from_csr(v, c, r).gauss_seidel(b, 0.0, iters)
    ups = W * H * iters / rep.seconds
    return {"workload": "512x512 single-channel Poisson, lexicographic Gauss-Seidel (reference order), 100 iterations",
            "kernel": "k_lex_wg (time-skewed strips: 8 sweeps per pass on the 8 compute waves of a workgroup, a loader and a "
                      "storer wave, all passes in one launch per pass depth)", "ms": rep.seconds * 1e3,
            "pixel_updates_per_s": ups,
            "bytes_model": f"{LEX_WG_BYTES_PER_UPDATE:.2f} B per update at 8 sweeps per pass (b 76/62 x 8/8, x read 63/62 x 8/8, x write "
                           "8/8, edge values 2 x 16/62); PMC on 16384^2: 4.2 B (profiles/r02_pmc_*_lex_wg_16384.csv)",
            "frac": ups * LEX_WG_BYTES_PER_UPDATE / 1e9 / HBM_PEAK_GBS,
            "bound_note": "not a bandwidth-bound kernel: lock-step steps of ~0.2 us (LDS round trip + barrier, tools/step_bench.hip); at "
                          "this size 10 strips x 12 passes on a critical path of ~(H + 64 + 60 strips) steps per pass",
            "bit_identical_to_oracle": bool(np.array_equal(x, want)),
            "cpu_baseline": {"value": W * H * iters / secs, "unit": "pixel-updates/s", "cores": 1, "kind": kind,
                             "sample": f"the same system and iteration count, {secs:.3f} s"}}


def config1(capi, cpu):
    """BASELINE configs[1]: 4096x4096 3-channel Poisson blend, red-black Gauss-Seidel."""
    W = H = 4096
    C, ips, steps = 3, 32, 10
    g = capi.Grid(W, H, C)
    g.randomize_x(1234, 0.0, 255.0)
    g.b_from_x()
    g.fill_x(1.0)
    T, R, _ = g.tune(8)                                         # (8, 140) on MI355X: the tiling the committed PMC traffic profile was made with
    g.sweep(ips)
    g.region_begin()
    for _ in range(steps):
        g.sweep(ips)
    ms, launches, _ = g.region_end()
    g.close()
    ups = float(W) * H * C * ips * steps / (ms * 1e-3)
    per_launch = ms / max(launches, 1)
    model = BYTES_PER_PIXEL_PASS * W * H * C
    return {"workload": "4096x4096 3-channel Poisson blend (one matrix, three right-hand sides), red-black Gauss-Seidel",
            "kernel": f"k_fused_sweep<{T},0,2>", "tiling": {"fused_depth": T, "rows_per_chunk": R, "tuned": True},
            "ms": ms / steps, "iters": ips, "pixel_updates_per_s": ups,
            "bytes_model": f"{BYTES_PER_PIXEL_PASS:.0f} B per pixel and channel per pass of {T} iterations",
            "avg_launch_ms": per_launch, "frac": model / (per_launch * 1e-3) / 1e9 / HBM_PEAK_GBS,
            **traffic_fields(f"{W}x{H}x{C}_T{T}_R{R}", per_launch * 1e-3),
            "frac_note": "x, b and the ping-pong buffer are 1.2 GB: past the 256 MiB Infinity Cache; the pass is bound by vector "
                         "issue (143.5 VALU instructions per march step, DESIGN section 4.1) at 23 % more marched rows per stored row "
                         "than the 16384^2 tiling (R = 140 + 32 against 364 + 32)",
            "cpu_baseline": cpu}


def config4(capi):
    """BASELINE configs[4]: 8192x8192 canvas, photomontage-style irregular mask, general CSR path."""
    import numpy as np
    from coursecomputationalphotography_amd import synth
    import oracle
    canvas, iters = 8192, 50
    mask = synth.disc_mask(canvas, canvas, seed=4321)
    v, c, r, colour, ys, xs = synth.masked_laplacian_csr(mask)
    n, nnz = len(ys), len(v)
    m = capi.CsrMatrix().upload_compressed(v, c, r)
    m.set_colouring(colour, 2)
    xt = synth.x_true(n, 4321)
    b = m.apply_to_vector(xt)
    m.gauss_seidel(b, 0.0, 2, check_every=0)                          # builds the schedule
    x, rep = m.gauss_seidel(b, 0.0, iters, check_every=0)
    rr, bb = m.residual_norm2(b, x)
    path = m.last_path()
    passes = m.last_sweep_launches
    # a count that is a whole number of depth-8 passes (the 50 above end in a pass of depth 2 that costs what a full one costs)
    _, rep64 = m.gauss_seidel(b, 0.0, 64, check_every=0)
    m.last_path()
    full = {"iters": 64, "passes": m.last_sweep_launches, "ms_per_iteration": rep64.seconds * 1e3 / 64, "row_updates_per_s": n * 64 / rep64.seconds}
    # the same matrix in the reference's own (index) order: 64 sweeps, and 8 of them against the C oracle
    m.gauss_seidel(b, 0.0, 8, check_every=0, ordering=capi.ORDER_LEXICOGRAPHIC)
    x_ref, rep_ref = m.gauss_seidel(b, 0.0, 64, check_every=0, ordering=capi.ORDER_LEXICOGRAPHIC)
    path_ref = m.last_path()
    x8, _ = m.gauss_seidel(b, 0.0, 8, check_every=0, ordering=capi.ORDER_LEXICOGRAPHIC)
    m.close()
    csr_bytes = 12.0 * nnz + 32.0 * n
    # the same matrix on the GENERAL path (what BASELINE configs[4] names: sliced-ELL images of the stored matrix,
    # colour-ordered sweep), the recognition switched off for this handle (CCP_GS_MASKED=0 is read at create)
    sell = None
    try:
        os.environ["CCP_GS_MASKED"] = "0"                                 # (read when the matrix is uploaded)
        ms_ = capi.CsrMatrix()
        ms_.upload_compressed(v, c, r)
        os.environ.pop("CCP_GS_MASKED", None)
        ms_.set_colouring(colour, 2)
        ms_.gauss_seidel(b, 0.0, 2, check_every=0)                        # builds the schedule
        xs, reps = ms_.gauss_seidel(b, 0.0, iters, check_every=0)
        sell = {"path": ms_.last_path(), "ms_per_iteration": reps.seconds * 1e3 / iters, "row_updates_per_s": n * iters / reps.seconds,
                "bytes_model": "SURVEY §8d CSR model: 12 B per stored entry + 32 B per row", "bytes_per_iteration": csr_bytes,
                "achieved_gbs": csr_bytes * iters / reps.seconds / 1e9, "frac": csr_bytes * iters / reps.seconds / 1e9 / HBM_PEAK_GBS,
                "same_bits_as_region_grid": bool(np.array_equal(xs, x)),
                **traffic_fields("sell_mask_8192", reps.seconds / (2 * iters))}
        ms_.close()
    except Exception as e:                                               # an extra must never cost the line
        sell = {"error": f"{type(e).__name__}: {e}"}
    finally:
        os.environ.pop("CCP_GS_MASKED", None)
    ups = n * iters / rep.seconds
    if path.startswith("region grid"):
        # recognised as the Laplacian of a raster region: swept matrix-free by the Dirichlet-mask grid.  Bytes an
        # unknown must move per PASS: x read 8 + b read 8 + x write 8 + mask 1 (halo re-reads, pixels of live tiles
        # outside the region and the canvas padding are the inefficiency the fraction shows)
        model = 25.0 * n * passes
        frac, model_txt = model / rep.seconds / 1e9 / HBM_PEAK_GBS, (f"25 B per unknown per pass (x 8 + b 8 + x write 8 + mask 1), {passes} passes "
                                                                      f"for the {iters} iterations")
    else:
        frac, model_txt = csr_bytes * iters / rep.seconds / 1e9 / HBM_PEAK_GBS, "SURVEY §8d CSR model: 12 B per stored entry + 32 B per row"
    om = oracle.Oracle().from_csr(v, c, r)
    t0 = time.perf_counter()
    want8 = om.gauss_seidel(b, 0.0, 8)[0]
    secs = time.perf_counter() - t0
    return {"workload": f"{canvas}x{canvas} canvas, union-of-discs + brush mask: {n} unknowns, {nnz} non-zeros, "
                        "5-point Laplacian restricted to the mask, 2-colour Gauss-Seidel",
            "path": path, "ms_per_iteration": rep.seconds * 1e3 / iters, "iters": iters, "passes": passes, "row_updates_per_s": ups,
            "whole_passes": full,
            "bytes_model": model_txt, "frac": frac,
            **(traffic_fields("region_grid_mask_8192", rep.seconds / max(passes, 1)) if path.startswith("region grid") else {}),
            "general_csr_path": sell,
            "csr_model_bytes_per_iteration": csr_bytes,
            "x_over_csr_streaming_roofline": csr_bytes * iters / rep.seconds / 1e9 / HBM_PEAK_GBS,
            "rel_residual_after": float(np.sqrt(rr / bb)),
            "reference_order": {"what": "the same matrix swept in index order (sparse-matrix.h:350-380 as it is), 64 sweeps",
                                "path": path_ref, "row_updates_per_s": n * 64 / rep_ref.seconds,
                                "bit_identical_to_oracle_8_sweeps": bool(np.array_equal(x8, want8))},
            "cpu_baseline": {"value": n * 8 / secs, "unit": "row-updates/s", "cores": 1, "kind": "port",
                             "sample": f"the same matrix, 8 lexicographic sweeps of the C oracle, {secs:.2f} s"}}


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

    from coursecomputationalphotography_amd import capi, rowblock

    W, H, C = args.width, args.height, args.channels
    parts = rowblock.partition_rows(H, world)
    row_begin, row_count = parts[rank]
    ghost = args.ghost if world > 1 else 0
    blk = rowblock.GridBlock(W, H, C, row_begin, row_count, ghost, local_rank)
    halo = args.halo if (world > 1 and args.backend == "nccl") else "torch"
    halo_note = None
    solver = None
    if halo == "abi":
        # the library's own RCCL communicator on every rank, or — decided together, every rank walking through the
        # same collectives whatever failed where — the torch.distributed exchange on every rank, so that a scaling
        # run still produces its line (rowblock_abi.setup_abi_solver)
        from coursecomputationalphotography_amd import rowblock_abi
        solver, why = rowblock_abi.setup_abi_solver(blk, rank, world, max(ghost, 2), dist, parts, H, overlap=not args.no_overlap)
        if solver is None:
            halo = "torch"
            halo_note = f"{why}: torch.distributed halo exchange used instead"
    if solver is None:
        solver = rowblock.make_solver(blk, rank, world, max(ghost, 2), dist, parts, H, halo="torch", overlap=False)
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
    per_call = ips if world == 1 else solver.iters_per_exchange
    tuned = False
    if args.depth > 0 and args.rows_per_chunk > 0:
        g.set_tiling(args.depth, args.rows_per_chunk)
    elif not args.tune and world == 1 and (W, H, C) in DEFAULT_TILING:
        g.set_tiling(*DEFAULT_TILING[(W, H, C)])
    else:
        # untimed: choose fused depth / chunk rows for this shape (speed only, results identical)
        g.tune(min(8, max(1, per_call // 2)))
        tuned = True
        g.synchronize()
    T, R, _ = g.get_tiling()
    for _ in range(args.warmup):
        solver.sweep(ips)
    barrier()
    g.region_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        solver.sweep(ips)
    barrier()
    elapsed = time.perf_counter() - t0
    region_ms, region_launches, region_iters = g.region_end()

    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    updates = float(W) * H * C * ips * args.steps
    value = updates / elapsed

    # roofline of the dominant kernel: HIP-event time of the timed region / the passes launched in it
    # (at N>1 the region also holds the halo exchanges; the last sweep call's own event pair is used instead)
    if world == 1:
        ms_per_launch = region_ms / max(region_launches, 1)
    else:
        ms, n_launch = g.last_timing()
        ms_per_launch = ms / max(n_launch, 1)
    depth = region_iters / max(region_launches, 1)            # what the planner really launched (DP over measured depths)
    roofline = roofline_of_pass(W, blk.local_rows, C, int(round(depth)), R, ms_per_launch, value, world)
    roofline["iterations_per_launch"] = depth
    roofline["launches_timed"] = int(region_launches)
    roofline["region_ms"] = region_ms

    extra = {}
    total_iters = (args.warmup + args.steps) * ips
    extra["rel_residual_after_timed"] = [total_iters, float(solver.rel_residual().max())]
    if world == 1 and not args.no_parity:
        extra["parity_check"] = parity_against_in_place(g, W, H, total_iters)
        extra["parity_check"]["tiling_checked"] = {"fused_depth": T, "rows_per_chunk": R}
    if not args.no_converge:
        if world == 1:
            k, trace = first_k_below(g, 1e-5, args.converge_cap)
            extra["iters_to_1e-5"] = k
            extra["rel_residual_trace"] = trace
            if k is not None:
                # the reference's own stop quantity, sum|x_k - x_{k-1}| (sparse-matrix.h:376), one more sweep
                extra["l1_step_after"] = [k + 1, float(g.sweep_l1().max())]
        else:
            # row-blocked: whole exchange intervals from a fresh start (granularity = one interval)
            g.fill_x(1.0)
            solver.exchange_halos()
            done, rel, trace = 0, float(solver.rel_residual().max()), []
            while rel > 1e-5 and done < args.converge_cap:
                solver.sweep(solver.iters_per_exchange)
                done += solver.iters_per_exchange
                rel = float(solver.rel_residual().max())
                trace.append([done, rel])
            extra["iters_to_1e-5"] = done if rel <= 1e-5 else None
            extra["iters_to_1e-5_granularity"] = solver.iters_per_exchange
            extra["rel_residual_trace"] = trace[-4:]

    if world == 1 and not args.no_reference_order:
        # untimed extra: the reference's OWN sweep order (lexicographic), bit-identical iterates, on the
        # same system from the same start vector (ccp_grid_gauss_seidel_lexicographic)
        g.fill_x(1.0)
        g.gauss_seidel_lexicographic(0.0, 4, 0)                       # allocations, code load
        g.fill_x(1.0)
        rep = g.gauss_seidel_lexicographic(0.0, args.reference_order_iters, 0)[0]
        extra["reference_order"] = {
            "what": "lexicographic Gauss-Seidel (the reference's index-order sweep, sparse-matrix.h:357-370), "
                    "time-skewed strips, 8 sweeps per pass (k_lex_wg); iterates bit-identical to the reference's",
            "iterations": rep.iterations, "seconds": rep.seconds,
            "pixel_updates_per_s": float(W) * H * C * rep.iterations / rep.seconds,
            "rel_residual_after": float(solver.rel_residual().max())}
        tr = load_traffic(f"lex_wg_{W}") if W == H and C == 1 else None
        if tr and tr.get("updates_per_launch"):
            bpu = tr["hbm_bytes_per_launch"] / tr["updates_per_launch"]
            ups = extra["reference_order"]["pixel_updates_per_s"]
            extra["reference_order"].update({"traffic_bytes_per_update": bpu, "traffic_source": tr["source"],
                                             "traffic_gbs": bpu * ups / 1e9, "traffic_frac_of_peak": bpu * ups / 1e9 / HBM_PEAK_GBS})

    if rank == 0:
        out = {
            "metric": "Gauss-Seidel pixel-updates/s on WxH 5-point Poisson grid",
            "value": value, "unit": "pixel-updates/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"{W}x{H} single-channel 5-point Poisson (SolveChannel closed form), "
                                    f"red-black Gauss-Seidel, fixed iteration count" if C == 1 else
                                    f"{W}x{H} {C}-channel Poisson blend, red-black Gauss-Seidel"),
                       "iters_per_step": ips, "channels": C,
                       "tiling": {"fused_depth": T, "rows_per_chunk": R, "tuned": tuned},
                       "partition": "single block" if world == 1 else solver.describe(),
                       **({"halo_note": halo_note} if halo_note else {})},
            "roofline": roofline,
        }
        out.update(extra)
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args.cpu_sample, args.cpu_iters)
            out["cpu_baseline"] = cpu
            out["vs_cpu_baseline"] = value / cpu["value"]
        if world == 1 and not args.no_configs:
            blk.close()
            cfg = {}
            for name, fn in (("configs[0]", lambda: config0(capi)),
                             ("configs[1]", lambda: config1(capi, out.get("cpu_baseline"))),
                             ("configs[4]", lambda: config4(capi))):
                try:
                    cfg[name] = fn()
                except Exception as e:                                # an extra must never cost the contract line
                    cfg[name] = {"error": f"{type(e).__name__}: {e}"}
            out["configs"] = cfg
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        solver.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
