#!/bin/bash
# round 3, batch 40: can the REAL RCCL run two ranks (two processes) on the one card of a test box?  If ncclCommInitRank
# accepts a duplicate GPU, the library's exchange runs over RCCL's own transport at world 2; if not, this records how it fails.
OUT=gpurun_out/r03
mkdir -p $OUT
make -C tests/cpp > /dev/null 2>&1
ID=$OUT/b40_id.bin
rm -f $ID
export NCCL_DEBUG=WARN
( timeout -k 5 120 ./tests/cpp/rowblock_driver 2 0 $ID 2048 1024 16 24 > $OUT/b40_rank0.log 2>&1; echo "rank0 rc=$?" >> $OUT/b40_rank0.log ) &
p0=$!
( timeout -k 5 120 ./tests/cpp/rowblock_driver 2 1 $ID 2048 1024 16 24 > $OUT/b40_rank1.log 2>&1; echo "rank1 rc=$?" >> $OUT/b40_rank1.log ) &
p1=$!
wait $p0 $p1
tail -8 $OUT/b40_rank0.log | cut -c1-300
tail -8 $OUT/b40_rank1.log | cut -c1-300
