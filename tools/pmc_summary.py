"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel: dispatches, average counter value.
FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE tallies 64 B per 128-B request, so
the byte figure for reads is doubled (MI355X_MICROARCH.md, confirmed by tools/hbm_calib on 8- and
16-byte-per-lane streaming reads: profiles/r01_hbm_calib.txt, r01_calib_*_counters.csv).

usage: pmc_summary.py <counter_collection.csv> <out.csv>
"""
import collections
import csv
import sys


def main(src, dst):
    agg = collections.OrderedDict()
    counters = []
    for r in csv.DictReader(open(src)):
        c = r["Counter_Name"]
        if c not in counters:
            counters.append(c)
        agg.setdefault((r["Kernel_Name"], r["Grid_Size"], c), []).append(float(r["Counter_Value"]))
    if len(counters) > 1:
        # several counters in one pass (SQ_*): one row per kernel, one column per counter (average per dispatch)
        kernels = collections.OrderedDict()
        for (k, gsz, c), v in agg.items():
            kernels.setdefault((k, gsz), {})[c] = (len(v), sum(v) / len(v))
        with open(dst, "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["kernel", "grid_size", "dispatches"] + [f"avg_{c}" for c in counters])
            for (k, gsz), d in kernels.items():
                w.writerow([k, gsz, max(n for n, _ in d.values())] + [f"{d[c][1]:.1f}" if c in d else "" for c in counters])
        return
    counter = counters[0] if counters else None
    scale = 2.0 if counter == "FETCH_SIZE" else 1.0
    with open(dst, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "grid_size", "dispatches", f"avg_{counter}_KB", "avg_bytes_corrected"])
        for (k, gsz, _), v in agg.items():
            avg = sum(v) / len(v)
            w.writerow([k, gsz, len(v), f"{avg:.3f}", f"{avg * 1024 * scale:.0f}"])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
