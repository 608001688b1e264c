#!/bin/bash
# Runs ON THE GPU BOX: k_ell_cls on an 8-way (and 4-way) shard of cfg3 -- together, the ELL blocks alone, the class workers alone
# (SVGPFA_ELLCLS_DIAG: wrong results, timing only), and with other numbers of ELL blocks per trial.
for w in 8 4; do
for d in 0 1 2; do echo "world $w diag $d"; SVGPFA_ELLCLS_DIAG=$d bash tests/tools/shard_trace.sh tr_diag$d $w | grep -E "ell_cls|period"; done
for p in 1 2 4; do echo "world $w ELL parts $p"; SVGPFA_ELLCLS_PARTS=$p bash tests/tools/shard_trace.sh tr_diagp$p $w | grep -E "ell_cls|period"; done
done
