#!/bin/bash
# Runs ON THE GPU BOX: tests/tools/shard_floor.py over a list of "LIB:ENV=VAL,ENV=VAL:worlds" cases (LIB = - for the default
# library, env list may be -), one line per (case, world).   usage: bash tests/tools/shard_matrix.sh case [case ...]
for c in "$@"; do
  IFS=: read lib envs worlds <<< "$c"
  ( [ "$lib" != "-" ] && export SVGPFA_LIB="$PWD/$lib"
    if [ "$envs" != "-" ]; then for kv in ${envs//,/ }; do export "$kv"; done; fi
    python tests/tools/shard_floor.py $worlds 2>/dev/null | grep " world " | sed "s|^|$c  |" )
done
