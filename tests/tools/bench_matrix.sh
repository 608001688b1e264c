#!/bin/bash
# Runs ON THE GPU BOX: bench.py over a list of "LIB:ENV=VAL,ENV=VAL:bench args" cases, one summary line each.
#   usage: bash tests/tools/bench_matrix.sh case [case ...]     (LIB = - for the default library, env list may be -)
for c in "$@"; do
  IFS=: read lib envs args <<< "$c"
  ( [ "$lib" != "-" ] && export SVGPFA_LIB="$PWD/$lib"
    if [ "$envs" != "-" ]; then for kv in ${envs//,/ }; do export "$kv"; done; fi
    python bench.py --no-cpu-baseline $args 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']; r2 = j.get('roofline_second') or {}
print('$c', round(j['value'], 2), round(j['ms_per_step'], 4), r['kernel'], round(r['kernel_avg_ms'], 4), r2.get('kernel'), round(r2.get('kernel_avg_ms', 0), 4))" )
done
