cd /tmp && export TMPDIR=/tmp
for m in 8192 100000000; do
  SVGPFA_MERGE_ITEMS=1000000 SVGPFA_MERGE_SPIKE_BLOCKS=$m timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ms$m -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bms$m.log 2>&1
  echo "merge_spike_blocks $m: $(grep -E '^\{' $GRAFT_REPO_ROOT/gpurun_out/bms$m.log | cut -c85-140)"
done
