#!/bin/bash
# round 3, batch 6: where the first solve of the 8192^2 mask spends its host time
OUT=gpurun_out/r03
mkdir -p $OUT
export TMPDIR=/tmp
CCP_GS_DEBUG=1 timeout -k 10 900 python tools/csr_bench.py > $OUT/b6_csr.json 2> $OUT/b6_csr.err; echo "rc=$?"
cat $OUT/b6_csr.json; grep "ccp_gs" $OUT/b6_csr.err | head -60
nproc; grep MemAvailable /proc/meminfo
