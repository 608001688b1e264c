for d in 0 1 2; do echo "diag $d"; SVGPFA_ELLCLS_DIAG=$d bash tests/tools/shard_trace.sh tr_diag$d 8 | grep -E "ell_cls"; done
for p in 1 2 4; do echo "parts $p (ELL alone)"; SVGPFA_ELLCLS_DIAG=1 SVGPFA_ELL_PARTS=$p bash tests/tools/shard_trace.sh tr_diagp$p 8 | grep -E "ell_cls"; done
echo "unmerged"; SVGPFA_ELL_CLS=0 bash tests/tools/shard_trace.sh tr_diagu 8 | grep -E "k_"
