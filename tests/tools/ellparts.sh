cd /tmp && export TMPDIR=/tmp
for p in 1 2 3 4 7; do
  SVGPFA_ELL_PARTS=$p timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ep$p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bep$p.log 2>&1
  echo "parts $p: $(grep -E '^\{' $GRAFT_REPO_ROOT/gpurun_out/bep$p.log | cut -c85-140)"
  grep quad_ell $GRAFT_REPO_ROOT/gpurun_out/prof_ep$p/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
done
