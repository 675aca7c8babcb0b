#!/bin/bash
# round 3, batch 3: exit probe (rocgdb), pass timelines, SQ counters of the fused pass, fused CG, whole GPU suite
OUT=gpurun_out/r03
mkdir -p $OUT
export TMPDIR=/tmp
echo "== exit probe"
timeout -k 10 900 python tools/exit_probe.py > $OUT/b3_exit_probe.log 2>&1; echo "rc=$?"
echo "== CG fused vs three passes"
for f in 1 0; do
  CCP_GS_CG_FUSED=$f timeout -k 10 300 python tools/cg_bench.py --size 8192 --height 4096 --channels 3 --iters 50 >> $OUT/b3_cg.jsonl 2>> $OUT/b3_cg.err
  CCP_GS_CG_FUSED=$f timeout -k 10 300 python tools/cg_bench.py --size 8192 --channels 1 --iters 50 >> $OUT/b3_cg.jsonl 2>> $OUT/b3_cg.err
done
cat $OUT/b3_cg.jsonl
echo "== pass traces"
for a in "4096 4096 3 8 140" "4096 4096 3 8 274" "4096 4096 3 8 342" "16384 16384 1 8 198 8192 2048 64" "16384 16384 1 8 364"; do
  timeout -k 10 300 python tools/pass_trace.py $a >> $OUT/b3_pass_trace.jsonl 2>> $OUT/b3_pass_trace.err || echo "trace $a failed"
  echo "trace $a done"
done
echo "== SQ counters of the fused pass"
lean="--no-configs --no-cpu-baseline --no-converge --no-parity --no-reference-order --steps 3 --warmup 1"
timeout -k 10 600 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVES -d $OUT/pmc_sq_bench -o bench -- python3 bench.py $lean > $OUT/b3_bench_under_sq.json 2> $OUT/b3_pmc_sq.log
echo "rc=$?"
f=$(find $OUT/pmc_sq_bench -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && python3 tools/pmc_summary.py "$f" $OUT/b3_pmc_sq_fused_16384.csv && cat $OUT/b3_pmc_sq_fused_16384.csv
timeout -k 10 600 rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM -d $OUT/pmc_sq2_bench -o bench -- python3 bench.py $lean > $OUT/b3_bench_under_sq2.json 2> $OUT/b3_pmc_sq2.log
echo "rc=$?"
f=$(find $OUT/pmc_sq2_bench -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && python3 tools/pmc_summary.py "$f" $OUT/b3_pmc_sq2_fused_16384.csv && cat $OUT/b3_pmc_sq2_fused_16384.csv
find $OUT -name '*counter_collection.csv' -delete
echo "== whole GPU suite"
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > $OUT/b3_all_tests.log 2>&1; echo "rc=$?"; tail -8 $OUT/b3_all_tests.log
