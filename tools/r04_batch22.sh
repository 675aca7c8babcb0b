#!/bin/bash
# round 4: whole GPU suite on the tree with the shifted strips (incl. the every-row 16384^2 oracle comparison)
mkdir -p gpurun_out/r04
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r04/tests_b22.log 2>&1
echo "tests rc=$?"; tail -14 gpurun_out/r04/tests_b22.log
