#!/bin/bash
# round 4: what a red-black pass costs at 512^2 (kernel trace: durations and gaps)
cat > /tmp/rb3.py <<'PY'
import os, sys, json; sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from coursecomputationalphotography_amd import capi
g = capi.Grid(512, 512, 1); g.randomize_x(1234, 0.0, 255.0); g.b_from_x()
g.fill_x(1.0); g.gauss_seidel(0.0, 16, 0)
g.fill_x(1.0); rep = g.gauss_seidel(0.0, 400, 0)[0]
print(rep.seconds)
PY
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof40
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof40 -o rb -- python3 /tmp/rb3.py 2>/dev/null | tail -1
f=$(find /tmp/prof40 -name "*kernel_stats.csv" | head -1)
python3 -c "import csv,sys; [print('  ', r['Name'][:70], r['Calls'], r['AverageNs']) for r in list(csv.DictReader(open(sys.argv[1])))[:5]]" $f
t=$(find /tmp/prof40 -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'fused' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=rows[-40:]
t0=int(rows[0]['Start_Timestamp'])
for r in rows[:16]:
    print(round((int(r['Start_Timestamp'])-t0)/1e3,1), round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,1), r['Kernel_Name'][10:40], r.get('Queue_Id'), r.get('Stream_Id'))
PY
