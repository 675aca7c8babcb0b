#!/bin/bash
# round 3, batch 4: multi-pass launch — parity, then A/B against single-pass launches; fused CG bit test; exit probe after the RTLD_LOCAL fix
OUT=gpurun_out/r03
mkdir -p $OUT
export TMPDIR=/tmp
echo "== multi parity + CG tests"
timeout -k 10 900 python -m pytest tests/test_gpu_multi.py tests/test_gpu_grid.py -x -q -m gpu > $OUT/b4_tests.log 2>&1; rc=$?; echo "rc=$rc"; tail -15 $OUT/b4_tests.log
[ $rc -ne 0 ] && exit 1
echo "== exit probe (RTLD_LOCAL)"
EXIT_PROBE_VARIANTS=rccl_by_path_then_torch,default_order timeout -k 10 600 python tools/exit_probe.py > $OUT/b4_exit_probe.log 2>&1; echo "rc=$?"; grep "=====" $OUT/b4_exit_probe.log
echo "== A/B"
for m in 0 1; do
  CCP_GS_MULTI=$m timeout -k 10 600 python tools/fused_ab.py big mid block >> $OUT/b4_ab.jsonl 2>> $OUT/b4_ab.err; echo "ab multi=$m rc=$?"
done
python - <<'PY'
import json
rows=[json.loads(l) for l in open('gpurun_out/r03/b4_ab.jsonl')]
for r in rows: print(r["multi"], r["case"], r["T"], r["R"], "%.4f ms"%r["ms_per_pass"], "%.3e"%r["updates_per_s"], "frac %.3f"%r["frac_24B"])
PY
echo "== whole GPU suite"
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > $OUT/b4_all_tests.log 2>&1; echo "rc=$?"; tail -8 $OUT/b4_all_tests.log
