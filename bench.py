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
TRAFFIC_FILES = [os.path.join(ROOT, "profiles", "r04_traffic.json"), os.path.join(ROOT, "profiles", "r03_traffic.json")]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=16384)
    ap.add_argument("--height", type=int, default=16384)
    ap.add_argument("--channels", type=int, default=1)
    ap.add_argument("--iters-per-step", type=int, default=32)
    ap.add_argument("--ghost", type=int, default=0,
                    help="ghost rows per side (N>1); exchange every ghost/2 iterations.  0 (default): timed on this node at set-up, "
                         "the fastest of 32 / 64 / 128 (rowblock_abi.choose_ghost)")
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
    ap.add_argument("--cpu-sample", type=int, default=16384,
                    help="edge of the CPU baseline grid: 16384 (default) is the headline system itself (BASELINE.md section 3), timed in a "
                         "child process beside the untimed part of the run after a 4096^2 sample that serves as the fallback")
    ap.add_argument("--cpu-iters", type=int, default=0, help="CPU baseline sweeps (default: 96 at 4096^2, 24 at 8192^2, 5 at 16384^2)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="host threads that BUILD the CPU baseline's system (the timed gaussSeidel is the "
                                                               "reference's serial sweep whatever this says); default min(32, cores)")
    ap.add_argument("--cpu-deadline", type=float, default=420.0, help="seconds the CPU baseline child may take before the largest finished sample is reported")
    ap.add_argument("--cpu-baseline-child", default="", help=argparse.SUPPRESS)     # internal: sizes, e.g. 4096,16384
    ap.add_argument("--cpu-log", default="", help=argparse.SUPPRESS)
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


def cpu_needs_wide_index(sample: int) -> bool:
    """The reference's default IndexType (int) sweeps a matrix only while every entry position stays below 2^30:
    getNearestIndex computes `(end + idx) / 2` in Index arithmetic (sparse-matrix.h:636).  SolveChannel's 16384^2 system
    has 1.34e9 entries — its last fifth of rows overflows (round 4's first attempt: > 400 s inside gaussSeidel, the
    search wandering over gigabytes, at(i, i) wrong) — so there the header is instantiated with a 64-bit IndexType."""
    return 5.0 * sample * sample > 2**30


def cpu_system_gb(sample: int) -> float:
    """Host memory of the CPU baseline at sample^2: CSR (8 B values + 4 or 8 B columns per entry, ~5 entries per row) twice
    (ours and the reference's ingest copy), its three row arrays, x_true, b and the reference's b copy, x and prev."""
    n = float(sample) * sample
    idx = 8 if cpu_needs_wide_index(sample) else 4
    return (2 * 5 * (8 + idx) * n + 4 * idx * n + 5 * 8 * n) / 2**30


def cpu_baseline_child(sizes, iters_arg: int, threads: int, log_path: str) -> None:
    """Runs in a process of its own (no torch, no GPU): for every size in turn, build the system, time the compiled
    reference's gaussSeidel on it and print ONE JSON line; every phase writes a wall-clock line to `log_path` as it ends,
    so a run that is cut short still says where the time went.

    Why phases matter (round 3 lost a 20-minute call here): on these virtual machines the FIRST touch of host memory the
    guest has never used costs 3-7 s per GB (the host backs guest pages lazily; measured in the build container: 1 GB in
    7.4 s on one thread, 16 GB in 49 s on eight, the same 16 GB in 4.9 s once the guest has had them before) — 45 GB of
    fresh pages for the 16384^2 system, and every numpy temporary on top.  So: (1) `warm` touches the memory the run
    will need on `threads` threads and frees it again, (2) the system is built band-wise by the C oracle's generator on
    those threads without temporaries, (3) the reference ingests it (its own serial copy, sparse-matrix.h:537-620, not
    timed) and (4) gaussSeidel (sparse-matrix.h:350-380, unmodified, serial by construction) is timed alone."""
    import numpy as np
    import oracle
    from concurrent.futures import ThreadPoolExecutor
    from coursecomputationalphotography_amd import synth
    t_start = time.perf_counter()
    log = open(log_path, "a") if log_path else sys.stderr

    def phase(sample, name, t0, extra=""):
        log.write(f"[cpu_baseline {time.perf_counter() - t_start:8.2f} s] {sample}^2 {name}: {time.perf_counter() - t0:.2f} s{extra}\n")
        log.flush()
        return time.perf_counter()

    orc = oracle.Oracle()
    try:
        ref = oracle.Ref()
    except (FileNotFoundError, OSError):
        ref = None
    for sample in sizes:
        need = cpu_system_gb(sample)
        avail = mem_available_gb()
        if avail < 1.2 * need + 2.0:
            log.write(f"[cpu_baseline] {sample}^2 skipped: needs {need:.0f} GB of host memory, {avail:.0f} GB available\n")
            log.flush()
            continue
        iters = iters_arg if iters_arg > 0 else (5 if sample > 8192 else 24 if sample > 4096 else 96)
        n = sample * sample
        phases = {}
        t0 = t_all = time.perf_counter()
        # (1) every page the run will use, touched once on many threads, then handed back to the allocator
        warm = np.empty(int(need * 2**30), dtype=np.uint8)
        step = -(-warm.size // threads)
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(lambda i: warm[i * step:(i + 1) * step:4096].fill(1), range(threads)))
        del warm
        t0 = phase(sample, "warm (first touch of host memory)", t0, f", {need:.1f} GB on {threads} threads")
        phases["warm"] = t0 - t_all
        wide = cpu_needs_wide_index(sample) and ref is not None
        v, c, r = orc.poisson_csr_threaded(sample, sample, threads, np.int64 if wide else np.int32)
        t1 = phase(sample, "generator (closed form, C oracle, row bands)", t0, f", {len(v)} entries, {'64' if wide else '32'}-bit indices")
        phases["generator"] = t1 - t0
        xt = synth.x_true(n, 1234)
        t2 = phase(sample, "x_true (mt19937, numpy)", t1)
        b = orc.poisson_apply_threaded(sample, sample, xt, threads)
        del xt
        t3 = phase(sample, "b = A x_true (row bands)", t2)
        phases["rhs"] = t3 - t1
        if ref is not None:
            kind = "reference"
            ingest, secs = (ref.gs_csr_timed_phases_i64 if wide else ref.gs_csr_timed_phases)(v, c, r, b, iters)
            phase(sample, "reference ingest (initializeFromEigenRowMajor, serial copy) + gaussSeidel", t3, f", of which ingest {ingest:.2f} s")
            phases["ingest"] = ingest
        else:
            kind = "port"
            m = orc.from_csr(v, c, r)
            t4 = phase(sample, "oracle ingest", t3)
            m.gauss_seidel(b, 0.0, iters)
            secs = time.perf_counter() - t4
        phase(sample, f"gaussSeidel, {iters} sweeps", time.perf_counter() - secs, f" = {n * iters / secs:.4g} pixel-updates/s")
        del v, c, r, b
        print(json.dumps({"sample": sample, "iters": iters, "kind": kind, "seconds_in_gauss_seidel": secs,
                          "value": float(n) * iters / secs, "phases_s": {k: round(x, 2) for k, x in phases.items()},
                          "host_gb": round(need, 1), "build_threads": threads,
                          "index_type": "int64" if wide else "int"}), flush=True)


class CpuBaselineRun:
    """The CPU baseline as a child process started after the timed region: it needs no GPU and minutes of host time
    (mostly first-touch page faults, see cpu_baseline_child), the untimed rest of the run needs the GPU and little host."""

    def __init__(self, args):
        import subprocess
        import threading
        sample = args.cpu_sample
        sizes = [4096] + ([sample] if sample != 4096 else [])
        self.threads = args.cpu_threads or max(1, min(32, os.cpu_count() or 1))
        out_dir = os.path.join(ROOT, "gpurun_out")
        self.log_path = args.cpu_log or (os.path.join(out_dir, "cpu_baseline_phases.log") if os.path.isdir(out_dir) else "")
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", ",".join(str(x) for x in sizes),
               "--cpu-iters", str(args.cpu_iters), "--cpu-threads", str(self.threads)]
        if self.log_path:
            cmd += ["--cpu-log", self.log_path]
        self.t0 = time.perf_counter()
        self.deadline = args.cpu_deadline
        self.lines = []
        self.proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
        self.reader = threading.Thread(target=lambda: [self.lines.append(ln) for ln in self.proc.stdout], daemon=True)
        self.reader.start()

    def result(self):
        import subprocess
        left = max(1.0, self.deadline - (time.perf_counter() - self.t0))
        note = ""
        try:
            self.proc.wait(timeout=left)
        except subprocess.TimeoutExpired:
            self.proc.kill()                                           # exactly the child started above
            self.proc.wait()
            note = f"; the child was stopped after {self.deadline:.0f} s (phases: {self.log_path or 'stderr'})"
        self.reader.join(timeout=5.0)
        done = []
        for ln in self.lines:
            try:
                done.append(json.loads(ln))
            except ValueError:
                pass
        if not done:
            return None
        best = done[-1]
        s = best["sample"]
        ph = best["phases_s"]
        what = ("the headline system itself" if s == 16384 else "a sample of the headline workload")
        wide_note = ("; SparseMatrix<double, IndexType = 64-bit>: with the default int the reference's getNearestIndex "
                     "(sparse-matrix.h:636, `(end + idx) / 2`) overflows beyond 2^30 stored entries and cannot sweep this system"
                     if best.get("index_type") == "int64" else "")
        return {"value": best["value"], "unit": "pixel-updates/s", "cores": 1, "kind": best["kind"],
                "host_cores_available": os.cpu_count(), "index_type": best.get("index_type"),
                "sample": f"{s}x{s} single-channel Poisson ({what}), {best['iters']} lexicographic iterations, "
                          f"{best['seconds_in_gauss_seidel']:.2f} s inside gaussSeidel (the sweep is serial by construction: 1 core); set-up on "
                          f"{best['build_threads']} host threads, not timed: first touch of {best['host_gb']} GB {ph.get('warm')} s, generator "
                          f"{ph.get('generator')} s, x_true and b {ph.get('rhs')} s, the reference's ingest copy {ph.get('ingest')} s{wide_note}{note}",
                "phases_s": ph, "smaller_samples": [{"sample": d["sample"], "value": d["value"], "iters": d["iters"]} for d in done[:-1]]}


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


def oracle_bands_check(g, W, H, iters=16, band=64):
    """The timed tiling against the reference-pinned ORACLE, pixel for pixel, on three bands of rows of this very system
    (top, middle, bottom): `iters` red-black iterations from x0 = 1 depend on 2 rows per iteration, so a band equals the
    oracle's run on the band plus 2*iters rows either side, cut out of the image with the same b (the oracle's int32
    positions — the reference's — cannot hold the whole 16384^2 system).  Checker only: never inside the timed region."""
    import numpy as np
    import oracle
    from coursecomputationalphotography_amd import synth
    orc = oracle.Oracle()
    g.fill_x(1.0)
    g.sweep(iters)
    pad, rows = 2 * iters + 2, []
    ok = True
    for y0 in (0, H // 2 - band // 2, H - band):
        lo, hi = max(0, y0 - pad), min(H, y0 + band + pad)
        lo -= lo & 1
        hs = hi - lo
        b = g.get_b(0, lo, hs).ravel()
        v, c, r = synth.poisson_csr(W, hs)
        want, _, _ = orc.multicolour_gauss_seidel(v, c, r, oracle.grid_colour(W, hs), b, 0.0, iters)
        same = bool(np.array_equal(g.get_x(0, y0, band), want.reshape(hs, W)[y0 - lo:y0 - lo + band]))
        ok &= same
        rows.append([y0, y0 + band])
    return {"against": "oracle/ccp_oracle.c (pinned to the compiled reference header) on three bands of rows of the timed system",
            "iterations": iters, "rows": rows, "bit_identical": ok}


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
    g.gauss_seidel_lexicographic(0.0, 8, 0)
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
                      "storer wave; persistent workgroups, all 100 sweeps one pipeline: the last group passes 4 sweeps through)",
            "ms": rep.seconds * 1e3,
            "pixel_updates_per_s": ups,
            "bytes_model": f"{LEX_WG_BYTES_PER_UPDATE:.2f} B per update at 8 sweeps per pass (b 76/62 x 8/8, x read 63/62 x 8/8, x write "
                           "8/8, edge values 2 x 16/62); PMC on 16384^2: 4.25 B (profiles/r04_pmc_*_lex.csv)",
            "frac": ups * LEX_WG_BYTES_PER_UPDATE / 1e9 / HBM_PEAK_GBS,
            "bound_note": "not a bandwidth-bound kernel: lock-step steps of ~0.23 us; at this size 13 groups of 9-12 strips on a critical "
                          "path set by the lag between neighbouring strips (31 us: 62 diagonals of geometry + 32 steps of hand-off) and "
                          "between groups (16 us): profiles/r04_lex_trace_512.jsonl",
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
            "frac_note": "geometry, not the kernel's speed (DESIGN section 4.1): per wave and march step the pass runs as fast as the "
                         "16384^2 one, but a chunk marches (R + 32) / R rows per stored row and the waves fill a fractional number of "
                         "rounds of the 2,048 wave slots",
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
    want8 = oracle.Oracle().from_csr(v, c, r).gauss_seidel(b, 0.0, 8)[0]         # the checker of x8 above
    kind, secs = cpu_gs_timed(v, c, r, b, 8)                                       # the baseline: the compiled reference where built
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
            "cpu_baseline": {"value": n * 8 / secs, "unit": "row-updates/s", "cores": 1, "kind": kind,
                             "sample": f"the same matrix, 8 lexicographic sweeps of "
                                       f"{'the compiled reference gaussSeidel' if kind == 'reference' else 'the C oracle'}, {secs:.2f} s"}}


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N rank processes ourselves (fresh interpreters, one per
    GPU, rendezvous on 127.0.0.1) BEFORE anything in this process touches the GPU, wait for them and return the worst
    exit code.  An external launcher (torch.distributed.run sets WORLD_SIZE) is honoured instead."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env))
    rc, live = 0, list(procs)
    while live:
        time.sleep(0.2)
        for pr in list(live):
            code = pr.poll()
            if code is None:
                continue
            live.remove(pr)
            if code != 0 and rc == 0:
                rc = code
                # a rank that failed leaves the others inside a collective: stop them (these exact children)
                deadline = time.time() + 20.0
                while live and time.time() < deadline:
                    time.sleep(0.2)
                    live = [q for q in live if q.poll() is None]
                for q in live:
                    q.kill()
    for pr in procs:
        pr.wait()
    return rc


def main():
    args = parse()
    if args.cpu_baseline_child:
        cpu_baseline_child([int(x) for x in args.cpu_baseline_child.split(",")], args.cpu_iters,
                           args.cpu_threads or max(1, min(32, os.cpu_count() or 1)), args.cpu_log)
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus))
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s)")
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
    use_abi = args.halo == "abi" and world > 1 and args.backend == "nccl"
    halo_note = None
    comm_box = {"comm": None, "why": None}

    def make_block(ghost_rows):
        """This rank's block with the synthetic system on it: x_true -> b = A x_true -> x0 = 1.0, all on device."""
        blk_ = rowblock.GridBlock(W, H, C, row_begin, row_count, ghost_rows if world > 1 else 0, local_rank)
        g_ = blk_.grid
        g_.randomize_x(1234, 0.0, 255.0)
        g_.b_from_x()
        g_.fill_x(1.0)
        return blk_

    def make_solver(blk_, ghost_rows):
        """The library's own RCCL communicator on every rank, or — decided together, every rank walking through the
        same collectives whatever failed where — the torch.distributed exchange on every rank, so that a scaling run
        still produces its line (rowblock_abi.setup_abi_solver).  One communicator serves every block of the run."""
        if use_abi and comm_box["why"] is None:
            from coursecomputationalphotography_amd import rowblock_abi
            sv, why = rowblock_abi.setup_abi_solver(blk_, rank, world, max(ghost_rows, 2), dist, parts, H,
                                                    overlap=not args.no_overlap, comm=comm_box["comm"])
            if sv is not None:
                if comm_box["comm"] is None:
                    comm_box["comm"] = sv.comm
                    sv.owns_comm = False                                   # closed at the end of the run, after every block
                return sv
            comm_box["why"] = why
        return rowblock.make_solver(blk_, rank, world, max(ghost_rows, 2), dist, parts, H, halo="torch", overlap=False)

    ghost, ghost_table = (args.ghost if world > 1 else 0), None
    if world > 1 and ghost <= 0:
        # ghost depth from what an exchange and an interval cost on THIS node (fewer, larger messages and more redundant
        # rows against more, smaller ones): every candidate timed on the real blocks, the decision all-reduced
        from coursecomputationalphotography_amd import rowblock_abi

        def tuned_block(gh):
            b_ = make_block(gh)
            b_.grid.tune(8)
            return b_
        ghost, ghost_table = rowblock_abi.choose_ghost(tuned_block, make_solver, dist, world)
    blk = make_block(ghost)
    solver = make_solver(blk, ghost)
    if use_abi and comm_box["why"] is not None:
        halo_note = f"{comm_box['why']}: torch.distributed halo exchange used instead"
    g = blk.grid
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
    # the CPU baseline needs minutes of host time and no GPU: a child process beside the untimed rest of the run
    cpu_run = CpuBaselineRun(args) if (world == 1 and rank == 0 and not args.no_cpu_baseline) else None

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
    exchange_cost = {}
    if world > 1:
        # untimed: one halo exchange alone, and one exchange interval (its sweeps + the exchange beside or after the
        # last pass), MAX over ranks — what the overlap has to hide and what is left of it
        def timed(fn, reps):
            barrier()
            t_ = time.perf_counter()
            for _ in range(reps):
                fn()
            g.synchronize()
            tt = torch.tensor([(time.perf_counter() - t_) / reps], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item()) * 1e3
        ipe = solver.iters_per_exchange
        solver.sweep(ipe)                                           # ends on an interval boundary whatever ips was
        exchange_cost = {"exchange_ms": timed(solver.exchange_halos, 8), "interval_ms": timed(lambda: solver.sweep(ipe), 8),
                         "interval_iterations": ipe}
    if world == 1 and not args.no_parity:
        extra["parity_check"] = parity_against_in_place(g, W, H, total_iters)
        extra["parity_check"]["tiling_checked"] = {"fused_depth": T, "rows_per_chunk": R}
        if H >= 256 and C == 1:
            try:
                extra["parity_check"]["oracle_bands"] = oracle_bands_check(g, W, H)
            except Exception as e:                                    # an extra must never cost the contract line
                extra["parity_check"]["oracle_bands"] = {"error": f"{type(e).__name__}: {e}"}
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
        g.gauss_seidel_lexicographic(0.0, 8, 0)                       # allocations, code load
        g.fill_x(1.0)
        rep = g.gauss_seidel_lexicographic(0.0, args.reference_order_iters, 0)[0]
        extra["reference_order"] = {
            "what": "lexicographic Gauss-Seidel (the reference's index-order sweep, sparse-matrix.h:357-370), "
                    "time-skewed strips, 8 sweeps per pass, persistent workgroups (k_lex_wg); iterates bit-identical to the reference's",
            "iterations": rep.iterations, "seconds": rep.seconds,
            "pixel_updates_per_s": float(W) * H * C * rep.iterations / rep.seconds,
            "rel_residual_after": float(solver.rel_residual().max())}
        tr = load_traffic(f"lex_wg_{W}") if W == H and C == 1 else None
        if tr and tr.get("updates_per_launch"):
            bpu = tr["hbm_bytes_per_launch"] / tr["updates_per_launch"]
            ups = extra["reference_order"]["pixel_updates_per_s"]
            extra["reference_order"].update({"traffic_bytes_per_update": bpu, "traffic_source": tr["source"],
                                             "traffic_gbs": bpu * ups / 1e9, "traffic_frac_of_peak": bpu * ups / 1e9 / HBM_PEAK_GBS})
        # the same at twice the sweeps: the three layout conversions (3.2 ms) and the pipeline's ramp weigh half as much
        g.fill_x(1.0)
        rep2 = g.gauss_seidel_lexicographic(0.0, 2 * args.reference_order_iters, 0)[0]
        extra["reference_order"]["at_twice_the_sweeps"] = {
            "iterations": rep2.iterations, "seconds": rep2.seconds,
            "pixel_updates_per_s": float(W) * H * C * rep2.iterations / rep2.seconds}
        # the reference's call as it stands: gaussSeidel(b) = epsilon 1e-6, at most 1,000 sweeps, the L1 step looked at
        # after every sweep (sparse-matrix.h:349-380); on a system of this size the rule never fires before the cap
        g.fill_x(1.0)
        rep3 = g.gauss_seidel_lexicographic(1e-6, 1000, 1)[0]
        extra["reference_order"]["default_call"] = {
            "what": "gaussSeidel(b) with the reference's defaults: epsilon 1e-6, max_iteration 1000, stop rule after every sweep",
            "iterations": rep3.iterations, "converged": rep3.converged, "seconds": rep3.seconds,
            "pixel_updates_per_s": float(W) * H * C * rep3.iterations / rep3.seconds}

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
        if world > 1:
            # what the multi-rank run really was: the ranks the library's communicator holds, the ghost depth chosen
            # (with the timings that chose it) and what one halo exchange costs beside one exchange interval
            info = comm_box["comm"].info() if comm_box["comm"] is not None else None
            out["multi_gpu"] = {"rccl_ranks_seen": info["world"] if info else None,
                                "rccl_version": info["rccl_version"] if info else None,
                                "halo_path": "libccp_gs.so over its own RCCL communicator" if info else f"torch.distributed ({args.backend})",
                                "ghost": ghost, "iters_per_exchange": solver.iters_per_exchange,
                                "ghost_candidates_ms_per_iteration": ghost_table, **exchange_cost}
        if world == 1 and not args.no_configs:
            blk.close()
            cfg = {}
            cfg_cpu = {"note": "see cpu_baseline of the line (one CPU run per bench run)"}
            for name, fn in (("configs[0]", lambda: config0(capi)),
                             ("configs[1]", lambda: config1(capi, cfg_cpu)),
                             ("configs[4]", lambda: config4(capi))):
                try:
                    cfg[name] = fn()
                except Exception as e:                                # an extra must never cost the contract line
                    cfg[name] = {"error": f"{type(e).__name__}: {e}"}
            out["configs"] = cfg
        if cpu_run is not None:
            cpu = cpu_run.result()
            if cpu is not None:
                out["cpu_baseline"] = cpu
                out["vs_cpu_baseline"] = value / cpu["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        solver.close()
        if comm_box["comm"] is not None:
            comm_box["comm"].close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
