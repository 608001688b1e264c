#!/bin/bash
# Runs ON THE GPU BOX: kernel trace of one rank's shard of cfg3 at WORLD-way sharding and the timeline of one step.
#   usage: bash tests/tools/shard_trace.sh <out tag> <world>
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/${1:-shard}
world=${2:-8}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace$world" -- python3 $root/tests/tools/shard_one.py $world 60 > "$out/trace$world.log" 2>&1 || exit 1
f=$(find "$out/trace$world" -name "*kernel_trace.csv" | head -1)
test -n "$f" || exit 1
python3 $root/tests/tools/step_timeline.py "$f" 10 > "$out/timeline$world.txt" 2>&1
find "$out/trace$world" -name "*kernel_trace.csv" -delete
cat "$out/timeline$world.txt"
