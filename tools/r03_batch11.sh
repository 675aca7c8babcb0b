#!/bin/bash
# round 3, batch 11: CSR row blocks at full size (configs[4]) — one block's kernel rate through the real RCCL at world 1,
# all 8 blocks as threads over the test transport (parity + message sizes), the new world-1 tests.
mkdir -p gpurun_out/r03
O=gpurun_out/r03
echo "== world-1 real RCCL tests"
timeout -k 10 300 python -m pytest tests/test_gpu_rccl.py -x -q > $O/b11_tests.log 2>&1; echo "rc=$?"; tail -3 $O/b11_tests.log
echo "== one block of 8 (real RCCL, world 1)"
timeout -k 10 500 python tools/csr_rows_bench.py block 8 > $O/b11_block.jsonl 2> $O/b11_block.err; echo "rc=$?"; cat $O/b11_block.jsonl; tail -3 $O/b11_block.err
echo "== 8 blocks as threads on one card (test transport)"
make -C tests/cpp > /dev/null 2>&1
CCP_GS_RCCL_LIB=$PWD/tests/cpp/libfake_rccl.so FAKE_RCCL_TIMEOUT_S=300 timeout -k 10 600 python tools/csr_rows_bench.py threads 8 > $O/b11_threads.jsonl 2> $O/b11_threads.err; echo "rc=$?"; cat $O/b11_threads.jsonl; tail -3 $O/b11_threads.err
