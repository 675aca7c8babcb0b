#!/bin/bash
# rocprofv3 passes for the kernels added in round 2 (run on the GPU box from the repo root): kernel trace of the
# strip-wave reference-order sweep, the Dirichlet-mask sweep and the edge hand-off pass; FETCH/WRITE PMC of the
# mask sweep.  Summaries under gpurun_out/$1/.
set -o pipefail
tag=${1:-r02_new}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --stats -d "$out/kt" -o k -- python3 tools/profile_kernels.py lex region edge > "$out/kt.txt" 2> "$out/kt.log" || exit 1
echo "kt done"
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --output-format csv --pmc $c -d "$out/pmc_${c}_region" -o k -- python3 tools/profile_kernels.py region > "$out/pmc_${c}_region.txt" 2> "$out/pmc_${c}_region.log" \
        && python3 tools/pmc_summary.py "$(find "$out/pmc_${c}_region" -name '*counter_collection.csv' | head -1)" "$out/pmc_${c}_region.csv"
    echo "pmc $c region done rc=$?"
done
python3 - "$out" <<'PY'
import csv, sys, collections
out = sys.argv[1]
# per-dispatch view of the edge experiment: duration of the fused kernels by variant
rows = list(csv.DictReader(open(f"{out}/kt/k_kernel_trace.csv")))
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "k_fused_sweep<8" in n or "k_fused_border<8" in n:
        agg[(n.split("(")[0], r["Grid_Size"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(f"{out}/edge_pass_kernels.csv", "w") as fh:
    fh.write("kernel,grid_size,dispatches,avg_us,min_us,max_us\n")
    for (k, gsz), v in sorted(agg.items()):
        fh.write(f"\"{k}\",{gsz},{len(v)},{sum(v)/len(v):.1f},{min(v):.1f},{max(v):.1f}\n")
print(open(f"{out}/edge_pass_kernels.csv").read())
PY
find "$out" -name '*counter_collection.csv' -delete
find "$out" -name '*kernel_trace.csv' -size +5M -delete
