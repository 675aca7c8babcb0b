#!/usr/bin/env python3
"""Per-workgroup timeline of the reference-order sweep (k_lex_wg) from the stamps it writes under CCP_GS_TRACE_FILE:
ticket taken, first gate passed (the strips it depends on are far enough ahead), last step done, where it ran.

  python tools/lex_trace.py run  W H sweeps [out.bin]    # one traced solve (the trace syncs after the launch)
  python tools/lex_trace.py show out.bin                  # summary as JSON lines

What the summary answers: how long a workgroup lives per step it executes (the lock-step step in the real kernel, against
the 131-161 ns of tools/step_bench.hip), how long it sits between its ticket and its first step (a slot held by a
workgroup that cannot run yet), how far apart neighbouring strips start (the ramp), and how many workgroups are alive
over time (the chip has 512 slots at two workgroups per CU)."""
import json
import os
import struct
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
TAG = 0x4341525458454C


def run(W, H, sweeps, path):
    if os.path.exists(path):
        os.remove(path)
    os.environ["CCP_GS_TRACE_FILE"] = path
    sys.path.insert(0, ROOT)
    from coursecomputationalphotography_amd import capi
    g = capi.Grid(W, H, 1)
    g.randomize_x(1234, 0.0, 255.0)
    g.b_from_x()
    g.fill_x(1.0)
    g.gauss_seidel_lexicographic(0.0, 8, 0)                  # allocations, code load (traced too: the first record)
    g.fill_x(1.0)
    rep = g.gauss_seidel_lexicographic(0.0, sweeps, 0)[0]
    print(json.dumps({"W": W, "H": H, "sweeps": sweeps, "seconds_with_trace": rep.seconds,
                      "updates_per_s": W * H * sweeps / rep.seconds}), flush=True)
    g.close()


def records(path):
    with open(path, "rb") as fh:
        data = fh.read()
    off = 0
    while off + 64 <= len(data):
        head = struct.unpack_from("<8Q", data, off)
        off += 64
        n = head[6]
        body = np.frombuffer(data, dtype="<u8", count=n, offset=off).reshape(-1, 4)
        off += 8 * n
        if head[0] == TAG:
            yield {"T": head[1], "groups": head[2], "S": head[3], "C": head[4], "H": head[5], "W": head[7]}, body


def show(path):
    for meta, r in records(path):
        T, G, S, H = meta["T"], meta["groups"], meta["S"], meta["H"]
        t0, t1, t2, where = (r[:, k].astype(np.int64) for k in range(4))
        ok = (t0 > 0) & (t2 > 0)
        base = t0[ok].min()
        tick = 10.0                                           # wall_clock64: 100 MHz
        start, gate, end = (t0 - base) * tick * 1e-3, (t1 - base) * tick * 1e-3, (t2 - base) * tick * 1e-3   # us
        tk = (r[:, 3] >> 32).astype(np.int64)
        grp, s = tk // S, tk % S
        steps = 64 + H + 2 * (T - 1) + 8                      # per strip (blocks of 8)
        life = end - start
        wait = gate - start
        run_ns = (end - gate) * 1e3 / steps
        out = {**meta, "workgroups": int(ok.sum()), "launch_us": float(end[ok].max()),
               "steps_per_strip": int(steps),
               "ns_per_step_running": {q: float(np.percentile(run_ns[ok], p)) for q, p in (("p10", 10), ("median", 50), ("p90", 90))},
               "us_ticket_to_first_gate": {q: float(np.percentile(wait[ok], p)) for q, p in (("p10", 10), ("median", 50), ("p90", 90), ("max", 100))},
               "share_of_slot_time_waiting": float(wait[ok].sum() / life[ok].sum())}
        # stagger between neighbouring strips of one group (first-gate times), and between groups at strip 0
        g0 = np.argsort(tk)
        gate_by = np.full(G * S, np.nan)
        gate_by[tk[ok]] = gate[ok]
        gb = gate_by.reshape(G, S)
        if S > 1:
            d = np.diff(gb, axis=1)
            out["us_between_neighbouring_strips"] = {"median": float(np.nanmedian(d)), "p90": float(np.nanpercentile(d, 90))}
        if G > 1:
            d = np.diff(gb[:, 0])
            out["us_between_groups_at_strip_0"] = {"median": float(np.nanmedian(d)), "p90": float(np.nanpercentile(d, 90))}
        # workgroups alive over time
        ev = np.concatenate([np.stack([start[ok], np.ones(ok.sum())], 1), np.stack([end[ok], -np.ones(ok.sum())], 1)])
        ev = ev[np.argsort(ev[:, 0], kind="stable")]
        alive = np.cumsum(ev[:, 1])
        dt = np.diff(ev[:, 0], append=ev[-1, 0])
        out["mean_workgroups_alive"] = float((alive * dt).sum() / max(dt.sum(), 1e-9))
        run_ev = np.concatenate([np.stack([gate[ok], np.ones(ok.sum())], 1), np.stack([end[ok], -np.ones(ok.sum())], 1)])
        run_ev = run_ev[np.argsort(run_ev[:, 0], kind="stable")]
        running = np.cumsum(run_ev[:, 1])
        dtr = np.diff(run_ev[:, 0], append=run_ev[-1, 0])
        out["mean_workgroups_past_their_first_gate"] = float((running * dtr).sum() / max(float(end[ok].max()), 1e-9))
        out["max_workgroups_alive"] = int(alive.max())
        # alive workgroups sampled over the launch (20 points), and how many share a CU (HW_ID: cu 11:8, sh 12, se 15:13)
        grid_t = np.linspace(0.0, float(end[ok].max()), 21)[:-1]
        out["alive_over_time"] = [int(((start[ok] <= t) & (end[ok] > t)).sum()) for t in grid_t]
        out["running_over_time"] = [int(((gate[ok] <= t) & (end[ok] > t)).sum()) for t in grid_t]
        xcc = ((r[:, 3] >> 24) & 0xF).astype(np.int64)
        hw = (r[:, 3] & 0xFFFFFF).astype(np.int64)
        cu_key = xcc * 4096 + ((hw >> 8) & 0xFF)
        out["distinct_cus"] = int(len(np.unique(cu_key[ok])))
        mid = grid_t[len(grid_t) // 2]
        live = ok & (start <= mid) & (end > mid)
        per_cu = np.bincount(np.unique(cu_key[live], return_inverse=True)[1]) if live.any() else np.zeros(1, dtype=int)
        out["workgroups_per_cu_at_mid_launch"] = np.bincount(per_cu).tolist()
        out["workgroups_per_xcc"] = np.bincount(xcc[ok], minlength=8).tolist()
        # the ideal: every slot-holder stepping at the isolated step time
        out["updates_per_s_of_this_launch"] = float(meta["W"]) * H * T * G / (float(end[ok].max()) * 1e-6)
        print(json.dumps(out), flush=True)


def table(path):
    """the last record's first-gate and last-step times, microseconds from the first ticket, one line per group"""
    last = None
    for meta, r in records(path):
        last = (meta, r)
    meta, r = last
    G, S = meta["groups"], meta["S"]
    t0, t1, t2 = (r[:, k].astype(np.int64) for k in range(3))
    ok = (t0 > 0) & (t2 > 0)
    base = t0[ok].min()
    tk = (r[:, 3] >> 32).astype(np.int64)
    gate = np.full(G * S, -1.0)
    end = np.full(G * S, -1.0)
    gate[tk[ok]] = (t1[ok] - base) * 1e-2
    end[tk[ok]] = (t2[ok] - base) * 1e-2
    print(json.dumps(meta))
    for g in range(G):
        print("g%-3d gate " % g + " ".join("%7.1f" % v for v in gate[g * S:(g + 1) * S]))
        print("     end  " + " ".join("%7.1f" % v for v in end[g * S:(g + 1) * S]))


if __name__ == "__main__":
    if sys.argv[1] == "table":
        table(sys.argv[2])
    elif sys.argv[1] == "run":
        run(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5] if len(sys.argv) > 5 else "gpurun_out/lex_trace.bin")
    else:
        show(sys.argv[2])
