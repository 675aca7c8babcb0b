"""profiles/r03_traffic.json from the round's PMC summaries (tools/pmc_summary.py output): HBM bytes per launch of the
kernels bench.py quotes, FETCH_SIZE doubled per the gfx950 correction (already applied in the summaries' last column) +
WRITE_SIZE.  usage: make_traffic.py <dir with pmc_FETCH_SIZE_*.csv / pmc_WRITE_SIZE_*.csv> <out.json>"""
import csv
import json
import os
import sys

ROUND = os.environ.get("ROUND", "r03")


def rows(path):
    out = {}
    if not os.path.exists(path):
        return out
    for r in csv.DictReader(open(path)):
        out.setdefault(r["kernel"], []).append((int(r["dispatches"]), float(r["avg_bytes_corrected"]), r["grid_size"]))
    return out


def per_launch(d, tag, *needles, pick="most"):
    """Sum over the kernels whose name contains a needle of the average bytes per dispatch (the dispatch group with the
    most dispatches of each kernel: the timed launches, not the one-off tuning shapes)."""
    f, w = rows(os.path.join(d, f"pmc_FETCH_SIZE_{tag}.csv")), rows(os.path.join(d, f"pmc_WRITE_SIZE_{tag}.csv"))
    tot_f = tot_w = 0.0
    found = []
    for needle in needles:
        for table, which in ((f, "f"), (w, "w")):
            hits = [(k, v) for k, v in table.items() if needle in k]
            if not hits:
                continue
            k, v = hits[0]
            best = max(v, key=lambda t: t[0])
            if which == "f":
                tot_f += best[1]
                found.append(f"{k.split('(')[0]} x{best[0]}")
            else:
                tot_w += best[1]
    return tot_f, tot_w, found


def main(d, out_path):
    out = {}

    def put(key, tag, needles, note, extra=None):
        f, w, found = per_launch(d, tag, *needles)
        if f + w <= 0:
            return
        out[key] = {"hbm_bytes_per_launch": f + w, "fetch_bytes_corrected": f, "write_bytes": w, "kernels": found,
                    "source": f"profiles/{ROUND}_pmc_FETCH_SIZE_{tag}.csv (FETCH_SIZE x 2: gfx950 correction) + profiles/{ROUND}_pmc_WRITE_SIZE_{tag}.csv; "
                              f"separate --pmc passes, tools/profile_round{ROUND[-1]}.sh; {note}"}
        if extra:
            out[key].update(extra)

    put("16384x16384x1_T8_R364", "bench", ["k_fused_sweep<8, 0, 2, false>", "k_fused_border<8, 0, 2, false>"],
        "bench.py's timed passes, pinned tiling", {"iterations_per_launch": 8})
    put("4096x4096x3_T8_R140", "mid", ["k_fused_sweep<8, 0, 2, false>", "k_fused_border<8, 0, 2, false>"],
        "tools/profile_kernels.py mid: BASELINE configs[1], the tiling bench.py pins", {"iterations_per_launch": 8})
    put("region_grid_mask_8192", "region", ["k_fused_sweep_masked<8, 0, 2>"],
        "tools/profile_kernels.py region: BASELINE configs[4] recognised as a raster region, depth-8 passes", {"iterations_per_launch": 8})
    put("sell_mask_8192", "sell", ["k_sell_gs<false>"],
        "tools/profile_kernels.py gs: BASELINE configs[4] on the sliced-ELL images, one launch per colour")
    put("lex_wg_16384", "lex", ["k_lex_wg<8, false>"],
        "tools/profile_kernels.py lex --sweeps 64: the reference-order sweep, 8 sweeps per pass, one launch for all passes",
        {"updates_per_launch": 16384.0 * 16384.0 * 64})
    put("cg_fused_8192x4096", "cg", ["k_cg_apply_march<false>", "k_cg_residual"],
        "tools/profile_kernels.py cg: passes A and B of one fused conjugate-gradient iteration on one channel of 8192x4096")
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e9, 3) for k, v in out.items()}))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
