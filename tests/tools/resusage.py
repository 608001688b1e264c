"""Summarise hipcc -Rpass-analysis=kernel-resource-usage remarks (stdin) as one line per kernel."""
import re, sys, subprocess
txt = sys.stdin.read()
cur = None; rows = []
for line in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    for key in ("VGPRs", "AGPRs", "SGPRs", "VGPRs Spill", "SGPRs Spill", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]", "ScratchSize [bytes/lane]"):
        m = re.search(r"remark: .*?\s" + re.escape(key) + r": (\d+)", line)
        if m and cur is not None and key not in cur:
            cur[key] = int(m.group(1))
seen = set()
for r in rows:
    if r["name"] in seen: continue
    seen.add(r["name"])
    try:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    except Exception:
        name = r["name"]
    print("%-62s vgpr %3s agpr %3s sgpr %3s vspill %3s scratch %4s occ %s" % (
        name[:62], r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs"), r.get("VGPRs Spill"),
        r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]")))
