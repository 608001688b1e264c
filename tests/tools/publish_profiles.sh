#!/bin/bash
# In the build container: condense what collect_profiles.sh / round_numbers.sh / round4_extras.sh merged back under
# gpurun_out/ into the committed artefacts profiles/<name>_*.   usage: bash tests/tools/publish_profiles.sh <name, e.g. r04_v4>
set -e
cd "$(dirname "$0")/../.."
v=$1
python tests/tools/digest_profiles.py $v $v | tail -1
n=gpurun_out/${v}_numbers
for f in bench_all bench_cfg2 bench_cfg3_mixed bench_cfg5 bench_cfg5_mixed bench_estep bench_estep_cached; do grep '^{' $n/$f.json > profiles/${v}_$f.json; done
for f in em_cfg3 em_cfg3_steps em_config1 shard_floor; do grep -v -i "amdgpu\|warn" $n/$f.txt > profiles/${v}_$f.txt; done
e=gpurun_out/${v}_extras
for k in 2 4; do grep '^{' $e/bench_gloo_n$k.json > profiles/${v}_bench_gloo_single_device_n$k.json; done
grep '^{' $e/bench_spread.json > profiles/${v}_bench_theta_spread.json
cp $e/timeline8.txt profiles/${v}_shard8_timeline.txt
cp $e/timeline4.txt profiles/${v}_shard4_timeline.txt
cp $e/spread_trace/*/*kernel_stats.csv profiles/${v}_theta_spread_kernel_stats.csv
[ -d $e/cfg5_trace ] && cp $e/cfg5_trace/*/*kernel_stats.csv profiles/${v}_cfg5_mixed_kernel_stats.csv
ls profiles | grep -c "^${v}_"
