"""Timeline of one steady-state step from a rocprofv3 kernel trace (csv): every kernel's start relative to the step's
k_build_reg, its duration, the queue it ran on and the idle gap on the critical chain before it.
usage: python tests/tools/step_timeline.py <kernel_trace.csv> [step index from the end, default 10]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 10
idx = [i for i, r in enumerate(rows) if "k_build_reg" in r["Kernel_Name"]]
per = sorted(int(rows[idx[i + 1]]["Start_Timestamp"]) - int(rows[idx[i]]["Start_Timestamp"]) for i in range(len(idx) // 2, len(idx) - 1))
print(f"steps {len(idx)}, median period of the second half {per[len(per) // 2] / 1e3:.1f} us (min {per[0] / 1e3:.1f}, max {per[-1] / 1e3:.1f})")
a, b = idx[-back - 1], idx[-back]
t0 = int(rows[a]["Start_Timestamp"])
end = t0
busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f}  gap {(s - end) / 1e3:6.1f}  q{r['Queue_Id']}  grid {r.get('Grid_Size', '?'):>8} {r['Kernel_Name'][:60]}")
    busy += e - s
    end = max(end, e)
print(f"step period {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, sum of kernel durations {busy / 1e3:.1f} us")
