#!/bin/bash
# round 3, batch 15: why the ring form of the window (28 % fewer vector instructions) is slower — SQ and instruction-cache
# counters of k_fused_sweep<8> at 16384^2, default (shifted) build against libccp_gs_ring2.so
OUT=$PWD/gpurun_out/r03
mkdir -p $OUT
export TMPDIR=/tmp
lean="--no-configs --no-cpu-baseline --no-converge --no-parity --no-reference-order --steps 3 --warmup 1"
for lib in default ring2; do
  if [ $lib = default ]; then unset CCP_GS_LIB; else export CCP_GS_LIB=$PWD/coursecomputationalphotography_amd/lib/libccp_gs_$lib.so; fi
  for grp in "sq:SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVES" \
             "ic:SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" \
             "vm:GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"; do
    tag=${grp%%:*}; ctr=${grp#*:}
    rm -rf $OUT/pmc_b15
    timeout -k 10 300 rocprofv3 --output-format csv --pmc $ctr -d $OUT/pmc_b15 -o k -- python3 bench.py $lean > $OUT/b15_${lib}_$tag.json 2> $OUT/b15_${lib}_$tag.log
    echo "rc=$? $lib $tag"
    f=$(find $OUT/pmc_b15 -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python3 tools/pmc_summary.py "$f" $OUT/b15_pmc_${lib}_$tag.csv && grep -E "^kernel|k_fused_sweep<8" $OUT/b15_pmc_${lib}_$tag.csv
  done
done
rm -rf $OUT/pmc_b15
