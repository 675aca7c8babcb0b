#!/bin/bash
# round 4: the default bench on the tree with the shifted strips
mkdir -p gpurun_out/r04
rm -f gpurun_out/cpu_baseline_phases.log
( time timeout -k 10 900 python bench.py ) > gpurun_out/r04/bench_b24.json 2> gpurun_out/r04/bench_b24.err
echo "bench rc=$?"; tail -4 gpurun_out/r04/bench_b24.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_b24.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['cpu_baseline']['value'])
print('ref order', d.get('reference_order', {}))
for k,c in d.get('configs',{}).items() if isinstance(d.get('configs'),dict) else enumerate(d.get('configs',[])):
    print(k, {kk:c.get(kk) for kk in ('ms','pixel_updates_per_s','row_updates_per_s','frac')})
PY
