#!/bin/bash
# Runs ON THE GPU BOX: bench.py over a list of "LIB:TILE_NW:variant" cases, one summary line each.
#   usage: bash tests/tools/bench_matrix.sh out_dir case [case ...]     (LIB = - for the default library)
out=$1; shift
mkdir -p "$out"
for c in "$@"; do
  IFS=: read lib nw var <<< "$c"
  log="$out/b_${lib//\//_}_${nw}_${var}.log"
  ( [ "$lib" != "-" ] && export SVGPFA_LIB="$PWD/$lib"; [ "$nw" != "-" ] && export SVGPFA_TILE_NW=$nw
    python bench.py --no-cpu-baseline --variant "$var" > "$log" 2>&1 )
  grep "^{" "$log" | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']; r2 = j.get('roofline_second') or {}
print('$c', round(j['value'], 1), round(j['ms_per_step'], 4), r['kernel'], round(r['kernel_avg_ms'], 4), r2.get('kernel'), round(r2.get('kernel_avg_ms', 0), 4), j['elbo'])"
done
