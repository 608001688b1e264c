"""GPU idle time inside a kernel trace (rocprofv3 --kernel-trace csv): total busy / idle over the last FRACTION of the run,
and the idle gaps summed by the kernel that precedes them (who was the GPU waiting behind: a host decision, a read-back).
usage: python tests/tools/trace_gaps.py <kernel_trace.csv> [fraction of the run, from the end, default 0.4]"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
t_end = int(rows[-1]["End_Timestamp"]); t_beg = int(rows[0]["Start_Timestamp"])
cut = t_end - frac * (t_end - t_beg)
rows = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
short = lambda n: re.sub(r"\(.*", "", re.sub(r"^void ", "", n))[:48]
busy = 0; end = int(rows[0]["Start_Timestamp"]); gaps = collections.Counter(); cnt = collections.Counter(); dur = collections.Counter(); prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if prev is not None and s > end:
        gaps[(short(prev), short(r["Kernel_Name"]))] += s - end
        cnt[(short(prev), short(r["Kernel_Name"]))] += 1
    busy += max(0, e - max(s, end)); end = max(end, e); prev = r["Kernel_Name"]
    dur[short(r["Kernel_Name"])] += e - s
span = end - int(rows[0]["Start_Timestamp"])
print(f"window {span/1e6:.2f} ms, busy {busy/1e6:.2f} ms ({100*busy/span:.1f} %), idle {(span-busy)/1e6:.2f} ms, {len(rows)} kernels")
print("idle gaps by (kernel before -> kernel after):")
for k, v in gaps.most_common(14):
    print(f"  {v/1e3:9.1f} us in {cnt[k]:4d} gaps (avg {v/1e3/cnt[k]:6.1f})  {k[0]} -> {k[1]}")
print("kernel time:")
for k, v in dur.most_common(14):
    print(f"  {v/1e3:9.1f} us  {k}")
