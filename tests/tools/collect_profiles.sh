#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): kernel-trace statistics and four separate PMC passes of the bench
# workload.  Counter passes never share a run with a trace domain other than --kernel-trace
# (MI355X_MICROARCH.md, HBM / rocprofv3 section).  Outputs land in gpurun_out/profile_<tag>/.
#   usage: bash tests/tools/collect_profiles.sh <tag> [bench.py args]
set -e -o pipefail
tag=${1:-run}; shift || true
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/profile_$tag
mkdir -p "$out"
# content hash of the device code these counters describe (bench.py ignores a digest whose hash differs from its tree)
(cd "$root" && python3 -c "import bench; print(bench.kernel_source_sha())") > "$out/csrc_sha.txt"
cd /tmp && export TMPDIR=/tmp
common="$root/bench.py --steps 20 --warmup 3 --no-cpu-baseline $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 $common > "$out/trace.log" 2>&1
echo "trace done"
small="$root/bench.py --steps 4 --warmup 1 --no-cpu-baseline $*"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 $small > "$out/pmc_fetch.log" 2>&1
echo "fetch done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 $small > "$out/pmc_write.log" 2>&1
echo "write done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d "$out/pmc_sq" -- python3 $small > "$out/pmc_sq.log" 2>&1
echo "sq done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d "$out/pmc_sq2" -- python3 $small > "$out/pmc_sq2.log" 2>&1
echo "sq2 done"
# keep only the small per-kernel files (gpurun merges <= 64 MiB back)
find "$out" -name "*kernel_trace.csv" -size +20M -delete
