#!/bin/bash
# Runs ON THE GPU BOX: k_item_bwd's phases on one rank's shard of an N-way split of cfg3 (default 8), with the
# stop-after-phase diagnosis builds (see phase_times.sh).
root=${GRAFT_REPO_ROOT:-$(pwd)}
world=${2:-8}
out=$root/gpurun_out/${1:-phases_shard}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for st in 9 10 1 2 6 7 8 3 0; do
  lib=$root/svgpfa_amd/csrc/libdiag_stop$st.so
  [ $st = 0 ] && lib=$root/svgpfa_amd/csrc/libsvgpfa_hip.so
  [ -f "$lib" ] || continue
  export SVGPFA_LIB=$lib
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stop$st" -- python3 $root/tests/tools/shard_profile.py $world > "$out/stop$st.log" 2>&1 || exit 1
  find "$out/stop$st" -name "*kernel_trace.csv" -delete
  f=$(find "$out/stop$st" -name "*kernel_stats.csv" | head -1)
  echo "stop $st: $(grep k_item_bwd "$f" | awk -F, '{print $(NF-4)}') ns avg k_item_bwd"
done
