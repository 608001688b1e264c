#!/bin/bash
# usage: trace_libs.sh <outdir> lib1 lib2 ...   (runs on the GPU box)
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/svgpfa_amd/csrc/libdiag_$v.so; [ "$v" = base ] && lib=$GRAFT_REPO_ROOT/svgpfa_amd/csrc/libsvgpfa_hip.so
  SVGPFA_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$v -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $out/$v.log 2>&1
  SVGPFA_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${v}_s8 -- python3 $GRAFT_REPO_ROOT/tests/tools/shard_one.py 8 40 > $out/${v}_s8.log 2>&1
  find $out -name "*kernel_trace.csv" -delete
  echo done $v
done
