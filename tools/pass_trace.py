"""Where the time of ONE temporally blocked pass goes: per-wave start/end stamps (CCP_GS_TRACE_FILE, written by
k_fused_sweep / k_fused_border themselves) turned into a timeline — waves in flight over time, tile durations by kind,
how many CUs carry one or two workgroups, when the last tile of each kind starts.

  python tools/pass_trace.py W H C T R [row_begin rows ghost [edge]]      -> one JSON object on stdout
(edge = 1: the pass is issued as the edges-first pass of a row block — ccp_grid_sweep_edges_first)
"""
import json
import os
import struct
import sys
import tempfile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def parse(path):
    passes = []
    with open(path, "rb") as fh:
        data = fh.read()
    off = 0
    while off + 64 <= len(data):
        head = struct.unpack_from("<8Q", data, off)
        off += 64
        assert head[0] == 0x43435054524143
        n = head[6]
        words = struct.unpack_from(f"<{n}Q", data, off)
        off += 8 * n
        passes.append((head, words))
    return passes


def summarise(head, words, bins=24):
    _, T, gx, gy, gz, bgx, n, R = head
    recs = []
    for i in range(0, n, 4):
        t0, t1, hw, tile = words[i:i + 4]
        if t1 == 0:
            continue
        hwid, xcc = hw & 0xffffffff, (hw >> 32) & 0xf
        cu = (xcc, (hwid >> 13) & 7, (hwid >> 12) & 1, (hwid >> 8) & 0xf)
        recs.append((t0, t1, cu, (hwid >> 4) & 3, (tile >> 40) & 0xff, tile & 0xffff, (tile >> 16) & 0xffff, (tile >> 32) & 0xff))
    if not recs:
        return {"error": "no wave wrote a record"}
    t_min = min(r[0] for r in recs)
    t_max = max(r[1] for r in recs)
    span = (t_max - t_min) / 100.0                                   # us (100 MHz clock)
    kinds = {0: "ordinary", 1: "top/bottom chunk (border kernel)", 2: "side strip (border kernel)"}
    out = {"T": T, "rows_per_chunk": R, "grid": [gx, gy, gz], "border_grid_x": bgx, "waves": len(recs), "span_us": span, "by_kind": {}}
    for k, name in kinds.items():
        d = [(r[1] - r[0]) / 100.0 for r in recs if r[4] == k]
        if d:
            st = [(r[0] - t_min) / 100.0 for r in recs if r[4] == k]
            out["by_kind"][name] = {"waves": len(d), "dur_us_avg": sum(d) / len(d), "dur_us_min": min(d), "dur_us_max": max(d),
                                    "first_start_us": min(st), "last_start_us": max(st),
                                    "last_end_us": max((r[1] - t_min) / 100.0 for r in recs if r[4] == k)}
    # waves in flight per time bin; wave-time integral / (span x 2048 wave slots at 2 per SIMD)
    hist = [0.0] * bins
    w = (t_max - t_min) / bins
    for r in recs:
        for b in range(bins):
            lo, hi = t_min + b * w, t_min + (b + 1) * w
            ov = min(r[1], hi) - max(r[0], lo)
            if ov > 0:
                hist[b] += ov / w
    out["waves_in_flight_per_bin"] = [round(h, 1) for h in hist]
    out["wave_time_integral_us"] = sum((r[1] - r[0]) for r in recs) / 100.0
    cus = {}
    for r in recs:
        cus.setdefault(r[2], []).append(r)
    out["cus_used"] = len(cus)
    per_cu = sorted(len(v) for v in cus.values())
    out["waves_per_cu_min_med_max"] = [per_cu[0], per_cu[len(per_cu) // 2], per_cu[-1]]
    busy = []
    for v in cus.values():
        busy.append(sum(r[1] - r[0] for r in v) / 100.0 / span / 8.0)  # fraction of the CU's 8 wave slots (2 per SIMD) over the span
    busy.sort()
    out["cu_slot_utilisation_min_med_max"] = [round(busy[0], 3), round(busy[len(busy) // 2], 3), round(busy[-1], 3)]
    # the waves that end last: (kind, chunk, strip, start us, duration us)
    out["last_waves"] = [[kinds.get(r[4], r[4]), r[5], r[6], round((r[0] - t_min) / 100.0, 1), round((r[1] - r[0]) / 100.0, 1)]
                         for r in sorted(recs, key=lambda r: -r[1])[:8]]
    ends = sorted((r[1] - t_min) / 100.0 for r in recs)
    out["end_percentiles_us"] = {p: ends[min(len(ends) - 1, int(len(ends) * p / 100))] for p in (50, 75, 90, 95, 99, 100)}
    return out


def main():
    a = [int(v) for v in sys.argv[1:]]
    W, H, C, T, R = a[:5]
    rb, rows, ghost = (a[5:8] + [0, 0, 0])[:3] if len(a) > 5 else (0, None, 0)
    edge = len(a) > 8 and a[8] != 0
    path = tempfile.mktemp(suffix=".trace")
    os.environ["CCP_GS_TRACE_FILE"] = path
    from coursecomputationalphotography_amd import capi
    g = capi.Grid(W, H, C, rb, rows or None, ghost, 0)
    g.randomize_x(1234, 0.0, 255.0)
    g.b_from_x()
    g.fill_x(1.0)
    g.set_tiling(T, R)
    if edge:
        g.sweep_edges_first(2 * T, ghost)
        g.halo_refreshed()
        g.sweep_edges_first(2 * T, ghost)
    else:
        g.sweep(2 * T)
        g.halo_refreshed()
        g.sweep(2 * T)
    g.synchronize()
    g.close()
    passes = parse(path)
    os.unlink(path)
    res = summarise(*passes[-1])
    res["shape"] = [W, H, C, rb, rows, ghost]
    res["edges_first"] = bool(edge)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
