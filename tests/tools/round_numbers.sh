#!/bin/bash
# Runs ON THE GPU BOX: the numbers README.md / DESIGN.md quote (bench variants, shard floor, EM wall-clock).
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/${1:-numbers}
mkdir -p "$out"
cd "$root"
python bench.py > "$out/bench_all.json" 2> "$out/bench_all.err"
python bench.py --variant estep --no-cpu-baseline > "$out/bench_estep.json" 2>/dev/null
python bench.py --variant estep --estep-cache --no-cpu-baseline > "$out/bench_estep_cached.json" 2>/dev/null
python bench.py --config cfg2 --no-cpu-baseline > "$out/bench_cfg2.json" 2>/dev/null
python bench.py --dtype mixed --no-cpu-baseline > "$out/bench_cfg3_mixed.json" 2>/dev/null
python bench.py --config cfg5 --steps 10 --warmup 2 --no-cpu-baseline > "$out/bench_cfg5.json" 2>/dev/null
python bench.py --config cfg5 --dtype mixed --steps 10 --warmup 2 --no-cpu-baseline > "$out/bench_cfg5_mixed.json" 2>/dev/null
python tests/tools/shard_floor.py > "$out/shard_floor.txt" 2>&1
python tests/tools/shard_floor.py --config cfg5 --dtype mixed 2 4 8 >> "$out/shard_floor.txt" 2>&1
python tests/tools/em_cfg3.py > "$out/em_cfg3.txt" 2>&1
python tests/tools/em_config1.py > "$out/em_config1.txt" 2>&1
python tests/tools/em_cfg3_steps.py 2 > "$out/em_cfg3_steps.txt" 2>&1
for f in "$out"/bench_*.json; do grep '^{' "$f" | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$(basename $f)', round(j['value'],2), round(j['ms_per_step'],4), round(j['roofline']['kernel_avg_ms'],4), round(j['roofline']['frac'],3))"; done
grep " world " "$out/shard_floor.txt"; tail -2 "$out/em_cfg3.txt"; tail -3 "$out/em_config1.txt"; grep -v amdgpu "$out/em_cfg3_steps.txt"
