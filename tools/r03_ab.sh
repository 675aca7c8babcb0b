#!/bin/bash
# round 3: A/B of the fused pass — XCD-contiguous tile order on/off, march unroll 2 (default build) / 4
set -e
OUT=gpurun_out/r03
mkdir -p $OUT
python -m pytest tests/test_gpu_grid.py tests/test_gpu_region.py -x -q -m gpu > $OUT/ab_tests.log 2>&1 || { tail -30 $OUT/ab_tests.log; exit 1; }
tail -3 $OUT/ab_tests.log
for lib in default u4; do
  for xcd in 1 0; do
    if [ $lib = default ]; then unset CCP_GS_LIB; else export CCP_GS_LIB=$PWD/coursecomputationalphotography_amd/lib/libccp_gs_$lib.so; fi
    CCP_GS_XCD=$xcd python tools/fused_ab.py big mid block region >> $OUT/ab.jsonl 2>> $OUT/ab.err
    echo "done $lib xcd=$xcd"
  done
done
cat $OUT/ab.jsonl
