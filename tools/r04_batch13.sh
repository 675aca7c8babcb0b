#!/bin/bash
# round 4: the dominant pass under three tile orders — launch order, XCD-contiguous runs of the row-major tile list, of a
# band-major list — time (HIP events over the bench's timed passes) and FETCH_SIZE / WRITE_SIZE per launch
mkdir -p gpurun_out/r04/xcd
export TMPDIR=/tmp
lean="--no-configs --no-cpu-baseline --no-converge --no-reference-order"
for m in 0 1 2; do
  CCP_GS_XCD=$m python3 bench.py $lean --steps 20 --warmup 5 > gpurun_out/r04/xcd/bench_xcd$m.json 2> gpurun_out/r04/xcd/bench_xcd$m.err
  python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r04/xcd/bench_xcd$m.json') if l.startswith('{')][-1])
print('CCP_GS_XCD=$m', 'value', d['value'], 'ms_per_launch', d['roofline']['avg_launch_ms'], 'parity', d['parity_check']['abs_sum_equal'] and d['parity_check']['bands_equal'])"
  for c in FETCH_SIZE WRITE_SIZE; do
    CCP_GS_XCD=$m timeout -k 10 300 rocprofv3 --output-format csv --pmc $c -d gpurun_out/r04/xcd/pmc_${c}_$m -o k -- python3 bench.py $lean --no-parity --steps 3 --warmup 1 > /dev/null 2> gpurun_out/r04/xcd/pmc_${c}_$m.log \
      && python3 tools/pmc_summary.py "$(find gpurun_out/r04/xcd/pmc_${c}_$m -name '*counter_collection.csv' | head -1)" gpurun_out/r04/xcd/pmc_${c}_xcd$m.csv
    grep "k_fused_sweep<8, 0, 2, false>" gpurun_out/r04/xcd/pmc_${c}_xcd$m.csv | cut -c1-200
  done
done
find gpurun_out/r04/xcd -name '*counter_collection.csv' -delete
