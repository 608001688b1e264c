#!/bin/bash
# Diagnosis builds of the library (in this container): svgpfa_amd/csrc/libdiag_<name>.so with extra -D flags, the
# translation units compiled with the same per-unit flags as __graft_entry__.build().
#   usage: bash tests/tools/build_diag.sh name1 "-DSV_BWD_STOP=1" [name2 "-D..."] ...     (two builds run at a time)
cd "$(dirname "$0")/../.."
one() {
  name=$1; flags=$2; d=/tmp/build_diag_$name; mkdir -p $d
  base="/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC $flags -c"
  $base -mllvm -amdgpu-mfma-vgpr-form svgpfa_amd/csrc/svgpfa_api.hip -o $d/a.o 2> $d/a.log &
  $base svgpfa_amd/csrc/lbfgs_api.hip -o $d/b.o 2> $d/b.log &
  $base -mllvm -disable-machine-licm svgpfa_amd/csrc/spike_cls_api.hip -o $d/c.o 2> $d/c.log &
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o svgpfa_amd/csrc/libdiag_$name.so $d/a.o $d/b.o $d/c.o && echo "built libdiag_$name.so"
}
n=0
while [ $# -ge 2 ]; do
  one "$1" "$2" &
  shift 2
  n=$((n + 1))
  [ $((n % 2)) = 0 ] && wait
done
wait
