#!/bin/bash
# Runs ON THE GPU BOX: per-kernel times of bench.py with the stop-after-phase diagnosis builds of the library
# (svgpfa_amd/csrc/libdiag_stop<N>.so, built with -DSV_BWD_STOP=N), to price the phases of k_item_bwd.
#   usage: bash tests/tools/phase_times.sh <out tag> [bench.py args, e.g. --config cfg5 --variant estep]
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/${1:-phases}
shift || true
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for st in 9 10 1 2 6 7 8 3 0; do
  lib=$root/svgpfa_amd/csrc/libdiag_stop$st.so
  [ $st = 0 ] && lib=$root/svgpfa_amd/csrc/libsvgpfa_hip.so
  [ -f "$lib" ] || continue
  export SVGPFA_LIB=$lib
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stop$st" -- python3 $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > "$out/stop$st.log" 2>&1 || exit 1
  find "$out/stop$st" -name "*kernel_trace.csv" -delete
  f=$(find "$out/stop$st" -name "*kernel_stats.csv" | head -1)
  echo "stop $st: $(grep k_item_bwd "$f" | awk -F, '{print $(NF-4)}') ns avg k_item_bwd"
done
