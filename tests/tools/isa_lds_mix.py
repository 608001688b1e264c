"""LDS instruction mix (and MFMA count) per kernel of the last build's device assembly (build/svgpfa_isa/*.s):
which kernels read their MFMA operands with ds_read_b64 (64 banks, 32-lane groups) and which had reads merged into
ds_read2_b64 / ds_read_b128 (other banking: MI355X_MICROARCH.md, LDS section).  usage: isa_lds_mix.py [name-substring ...]"""
import collections
import glob
import os
import re
import sys

REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
pats = sys.argv[1:]
for path in sorted(glob.glob(os.path.join(REPO, "build", "svgpfa_isa", "*amdgcn*.s"))):
    txt = open(path).read()
    for m in re.finditer(r"^(_Z\w+):\s*;", txt, re.M):
        n = m.group(1)
        if pats and not any(p in n for p in pats):
            continue
        e = re.search(r"\.Lfunc_end\d+:", txt[m.start():])
        body = txt[m.start():m.start() + e.start()]
        c = collections.Counter(re.findall(r"^\s+(ds_\w+)", body, re.M))
        print(n[:60], dict(c), "mfma", len(re.findall(r"v_mfma", body)))
