cd /tmp && export TMPDIR=/tmp
for nw in 4 5 7 8; do
  SVGPFA_ELL_NW=$nw timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_nw$nw -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bnw$nw.log 2>&1
  echo "nw $nw: $(grep -E '^\{' $GRAFT_REPO_ROOT/gpurun_out/bnw$nw.log | cut -c85-140) ell $(grep quad_ell $GRAFT_REPO_ROOT/gpurun_out/prof_nw$nw/*/*kernel_stats.csv | awk -F'",' '{split($2,a,","); printf "%.1f", a[3]/1000}')"
done
