#!/bin/bash
# round 4: a longer randomised parity soak of the final build (new seeds), progress line per seed
mkdir -p gpurun_out/r04
for seed in ${SOAK_SEEDS:-11 12 13 14 15 16}; do
  timeout -k 10 280 python -c "
import sys; sys.path.insert(0,'tools'); import soak
bad = soak.run(${SOAK_N:-40}, $seed); print('seed', $seed, 'failures', bad, flush=True); sys.exit(1 if bad else 0)" >> gpurun_out/r04/soak.log 2>&1 || { echo "seed $seed: FAILED or timed out"; tail -5 gpurun_out/r04/soak.log; exit 1; }
  tail -1 gpurun_out/r04/soak.log
done
