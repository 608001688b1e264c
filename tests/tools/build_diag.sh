#!/bin/bash
# Diagnosis builds of the library (in this container): svgpfa_amd/csrc/libdiag_<name>.so with extra -D flags.
#   usage: bash tests/tools/build_diag.sh name1 "-DSV_BWD_STOP=1" [name2 "-D..."] ...     (up to four builds run at a time)
cd "$(dirname "$0")/../.."
n=0
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC $flags -o svgpfa_amd/csrc/libdiag_$name.so \
      svgpfa_amd/csrc/svgpfa_api.hip svgpfa_amd/csrc/lbfgs_api.hip svgpfa_amd/csrc/spike_cls_api.hip 2> /tmp/build_diag_$name.log && echo "built libdiag_$name.so" ) &
  n=$((n + 1))
  [ $((n % 4)) = 0 ] && wait
done
wait
