#!/bin/bash
# round 3, batch 7: device-side region recognition (parity + time), follow-up chunks of the edges-first pass
OUT=gpurun_out/r03
mkdir -p $OUT
export TMPDIR=/tmp
make -C tests/cpp > /dev/null 2>&1
echo "== parity: region / csr / insert / rccl (device recognition, follow-up chunks)"
timeout -k 10 1100 python -m pytest tests/test_gpu_region.py tests/test_gpu_csr.py tests/test_gpu_insert.py tests/test_gpu_rccl.py tests/test_gpu_rccl_multirank.py tests/test_gpu_fullsize.py -x -q -m gpu > $OUT/b7_tests.log 2>&1; rc=$?; echo "rc=$rc"; tail -12 $OUT/b7_tests.log
echo "== the same region tests with the host recognition"
CCP_GS_REGION_HOST=1 timeout -k 10 600 python -m pytest tests/test_gpu_region.py -x -q -m gpu > $OUT/b7_tests_host.log 2>&1; echo "rc=$?"; tail -3 $OUT/b7_tests_host.log
echo "== first solve of the 8192^2 mask"
CCP_GS_DEBUG=1 timeout -k 10 900 python tools/csr_bench.py > $OUT/b7_csr.json 2> $OUT/b7_csr.err; echo "rc=$?"
cat $OUT/b7_csr.json; grep "ccp_gs" $OUT/b7_csr.err | grep -v "tune T" | head -40
echo "== edges-first pass: timeline and interval time"
rm -f $OUT/b7_pass_trace.jsonl
for a in "16384 16384 1 8 198 8192 2048 64 0" "16384 16384 1 8 198 8192 2048 64 1"; do
  timeout -k 10 300 python tools/pass_trace.py $a >> $OUT/b7_pass_trace.jsonl 2>> $OUT/b7_pass_trace.err || echo "trace $a failed"
done
python - <<'PY'
import json
for l in open('gpurun_out/r03/b7_pass_trace.jsonl'):
    d=json.loads(l); print("edges_first", d["edges_first"], "R", d["rows_per_chunk"], "grid", d["grid"], "span_us %.1f"%d["span_us"], {k:(v["waves"], round(v["dur_us_avg"],1), round(v["dur_us_max"],1), round(v["last_end_us"],1)) for k,v in d["by_kind"].items()})
PY
for m in plain edges; do timeout -k 10 300 python tools/rank_block_bench.py 8 64 $m >> $OUT/b7_rank_block.jsonl 2>> $OUT/b7_rank_block.err; done
CCP_GS_EDGE_FOLLOW=0 timeout -k 10 300 python tools/rank_block_bench.py 8 64 edges >> $OUT/b7_rank_block.jsonl 2>> $OUT/b7_rank_block.err
cat $OUT/b7_rank_block.jsonl
