cd /tmp && export TMPDIR=/tmp
for cfg in "4 1" "2 1" "2 2" "3 1"; do
  set -- $cfg
  SVGPFA_ELL_NW=$1 SVGPFA_ELL_PARTS=$2 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_nw$1$2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bnw$1$2.log 2>&1
  echo "nw $1 parts $2: $(grep -E '^\{' $GRAFT_REPO_ROOT/gpurun_out/bnw$1$2.log | cut -c85-140) ell $(grep quad_ell $GRAFT_REPO_ROOT/gpurun_out/prof_nw$1$2/*/*kernel_stats.csv | awk -F'",' '{split($2,a,","); printf "%.1f", a[3]/1000}')"
done
