#!/bin/bash
# round 4: where the time between the events of a 256-sweep call goes (kernel trace: gaps between kernels)
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof26
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace -d /tmp/prof26 -o lex -- python3 $GRAFT_REPO_ROOT/tools/lex_grid_bench.py > $GRAFT_REPO_ROOT/gpurun_out/r04/lex_b26.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof26 -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
prev_end=None
out=[]
for r in rows:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    name=r['Kernel_Name'][:60]
    gap=(s-prev_end)/1e6 if prev_end else 0
    out.append((round((s-t0)/1e6,2),round((e-s)/1e6,3),round(gap,3),name))
    prev_end=e
# print the last 60 kernels (the 16384^2 x 256 part)
for o in out[-60:]: print(o)
PY
grep "^{" gpurun_out/r04/lex_b26.log | tail -1 | cut -c1-200
