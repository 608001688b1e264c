#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): one rocprofv3 kernel-trace of the bench workload, per-kernel averages to stdout.
#   usage: bash tests/tools/quick_trace.sh <tag> [bench.py args]
set -e -o pipefail
tag=${1:-quick}; shift || true
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/trace_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > "$out/trace.log" 2>&1
python3 - "$out" <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print("%-64s %5s %9.1f us" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3))
P
tail -1 "$out/trace.log" | cut -c1-200
find "$out" -name "*kernel_trace.csv" -size +20M -delete
