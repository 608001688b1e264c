#!/bin/bash
# Runs ON THE GPU BOX: k_ell_cls on an 8-way (and 4-way) shard of cfg3 -- together, the ELL blocks alone, the class workers alone
# (diagnosis BUILDS of the library, -DSV_ELLCLS_DIAG=1 / 2: wrong results, timing only; make them in the build container first:
#   bash tests/tools/build_diag.sh ellcls1 "-DSV_ELLCLS_DIAG=1" ellcls2 "-DSV_ELLCLS_DIAG=2"), and with other numbers of
# ELL blocks per trial.
root=${GRAFT_REPO_ROOT:-$(pwd)}
for w in 8 4; do
for d in 0 1 2; do
  lib=$root/svgpfa_amd/csrc/libdiag_ellcls$d.so
  [ $d = 0 ] && lib=$root/svgpfa_amd/csrc/libsvgpfa_hip.so
  [ -f "$lib" ] || continue
  echo "world $w diag $d"; SVGPFA_LIB=$lib bash tests/tools/shard_trace.sh tr_diag$d $w | grep -E "ell_cls|period"
done
for p in 1 2 4; do echo "world $w ELL parts $p"; SVGPFA_ELLCLS_PARTS=$p bash tests/tools/shard_trace.sh tr_diagp$p $w | grep -E "ell_cls|period"; done
done
