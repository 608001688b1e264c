#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel trace of two ECM iterations at cfg3 (em_cfg3_steps.py); prints, for the
# last E-step maximisation, the kernels of three consecutive closures with the gaps between them.
set -e -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/em_trace
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$out/trace" -- python3 $root/tests/tools/em_cfg3_steps.py 2 > "$out/trace.log" 2>&1
python3 - "$out" <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# E-step closures: evaluations whose k_item_bwd is the <*, false> variant
idx = [i for i, r in enumerate(rows) if "k_item_bwd<2, false>" in r["Kernel_Name"]]
sel = idx[-12:-8]
a, b = sel[0], sel[-1]
# back up to the k_front in front of the first one
while a > 0 and "k_front" not in rows[a]["Kernel_Name"]: a -= 1
t0 = int(rows[a]["Start_Timestamp"]); prev = t0
busy = 0
for r in rows[a:b + 2]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f dur %7.1f gap %6.1f  %s" % ((st - t0) / 1e3, (en - st) / 1e3, (st - prev) / 1e3, r["Kernel_Name"][:70]))
    busy += en - st; prev = en
print("span %.1f us, kernels %.1f us" % ((prev - t0) / 1e3, busy / 1e3))
P
find "$out" -name "*kernel_trace.csv" -size +30M -delete
