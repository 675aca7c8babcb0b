#!/bin/bash
# round 4 experiment: is k_lex_wg's step waiting for its loader's memory latency?  Same strip with the loader's prefetches
# (b, x or both) replaced by arithmetic (wrong results, same instruction stream otherwise) against the shipped build.
mkdir -p gpurun_out/r04
for lib in libccp_gs.so libccp_gs_nob.so libccp_gs_nox.so libccp_gs.so; do
  echo "== $lib"
  CCP_GS_LIB=$PWD/coursecomputationalphotography_amd/lib/$lib timeout -k 10 120 python tools/lex_trace.py run 16384 16384 64 gpurun_out/r04/trace_y.bin || exit 1
  python tools/lex_trace.py show gpurun_out/r04/trace_y.bin | tail -1 | cut -c1-420
done
