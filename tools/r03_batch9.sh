#!/bin/bash
# round 3, batch 9: sliced-ELL images built on the device — whole suite, then set-up times (general path, edits)
OUT=gpurun_out/r03
mkdir -p $OUT
export TMPDIR=/tmp
make -C tests/cpp > /dev/null 2>&1
echo "== whole GPU suite"
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $OUT/b9_all_tests.log 2>&1; rc=$?; echo "rc=$rc"; tail -12 $OUT/b9_all_tests.log
echo "== csr tests with host-built images (A/B of the same assertions)"
CCP_GS_SCHEDULE_HOST=1 timeout -k 10 600 python -m pytest tests/test_gpu_csr.py tests/test_gpu_insert.py -x -q -m gpu > $OUT/b9_tests_hostsched.log 2>&1; echo "rc=$?"; tail -3 $OUT/b9_tests_hostsched.log
echo "== set-up times at the 8192^2 mask"
CCP_GS_DEBUG=1 timeout -k 10 600 python tools/csr_bench.py > $OUT/b9_csr.json 2> $OUT/b9_csr.err; echo "rc=$?"
cat $OUT/b9_csr.json; grep "ccp_gs" $OUT/b9_csr.err | grep -v "tune T" | head -40
CCP_GS_DEBUG=1 timeout -k 10 900 python tools/insert_bench.py > $OUT/b9_insert.json 2> $OUT/b9_insert.err; echo "rc=$?"
cat $OUT/b9_insert.json; grep "ccp_gs" $OUT/b9_insert.err | grep -v "tune T" | head -60
