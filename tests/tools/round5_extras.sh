#!/bin/bash
# Runs ON THE GPU BOX: the round-5 extras -- timelines of one rank's 8- and 4-way shard step, the cfg3 bench on diverged
# kernel parameters (--theta-spread) with its kernel statistics, a kernel trace of cfg5 (--dtype mixed), and the N > 1 bench
# line with 2 and 4 ranks sharing cuda:0 over gloo, started by bench.py ITSELF (`python bench.py --gpus N`, no launcher).
#   usage: bash tests/tools/round5_extras.sh <out tag>
root=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r05_extras}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
bash tests/tools/shard_trace.sh $tag 8 > /dev/null 2>&1
bash tests/tools/shard_trace.sh $tag 4 > /dev/null 2>&1
python bench.py --theta-spread --no-cpu-baseline > "$out/bench_spread.json" 2> "$out/bench_spread.err"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/spread_trace" -- python3 $root/bench.py --theta-spread --steps 20 --warmup 3 --no-cpu-baseline > "$out/spread_trace.log" 2>&1)
find "$out/spread_trace" -name "*kernel_trace.csv" -delete
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/cfg5_trace" -- python3 $root/bench.py --config cfg5 --dtype mixed --steps 10 --warmup 2 --no-cpu-baseline > "$out/cfg5_trace.log" 2>&1)
find "$out/cfg5_trace" -name "*kernel_trace.csv" -delete
for n in 2 4; do
  HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 300 python bench.py --gpus $n --backend gloo --single-device --no-cpu-baseline > "$out/bench_gloo_n$n.json" 2> "$out/bench_gloo_n$n.err"
done
cat "$out/timeline8.txt" "$out/timeline4.txt"
grep -h '^{' "$out"/bench_spread.json "$out"/bench_gloo_n*.json | cut -c1-400
