"""Scan AMDGPU assembly (hipcc -S --cuda-device-only) for the ROCm 7.2 / clang 22 code-generation defect found in round 3
(DESIGN.md section 9): a register-allocator copy of a VGPR placed at the top of a control-flow JOIN block *in front of*
the `s_or_b64 exec, exec, s[..]` that re-enables the lanes of the other branch.  The copy then runs under the mask of one
branch only -- for a wave whose lanes all took the other branch, under exec = 0 -- and the copied value (in the case
found: threadIdx.x inside k_item_bwd<3, true>) is garbage afterwards.

A block is suspect when it is the target of an `s_cbranch_execz` (a join) and, between its label and its
`s_or_b64 exec, exec, ...`, it holds ANY instruction whose effect depends on EXEC -- a vector-register copy (the case
found), or any other VALU / LDS / memory instruction (round 4, after review: a spill store or an accumulator move in
another operand form in that place would be the same defect) -- except the ones that are legitimate there: scalar
instructions, SGPR spills (v_writelane / v_readlane ignore EXEC) and splats of an SGPR or a constant into a VGPR (every
lane gets the same value whatever the mask... for the lanes that are on; such a register is rematerialised, not saved).

    python tests/tools/isa_exec_guard.py file.s [...]      exit status 1 if any suspect block is found
"""
import re
import sys

LABEL = re.compile(r"^(\.LBB\d+_\d+|[A-Za-z_][\w$.]*):")
EXEC_RESTORE = re.compile(r"^\s*s_or_b64\s+exec,\s*exec,")
EXECZ_TARGET = re.compile(r"^\s*s_cbranch_execz\s+(\.LBB\d+_\d+)")
# copies: `v_mov dst, src` and the accumulator moves; harmless when the source is an SGPR or a literal (rematerialisation)
COPY = re.compile(r"^\s*(v_mov_b32_e32|v_mov_b64_e32|v_accvgpr_write_b32|v_accvgpr_read_b32|v_accvgpr_mov_b32)\s+([av]\[?[\d:]+\]?),\s*(\S+)\s*$")
VECTOR_SRC = re.compile(r"^[av]\[?[\d:]+\]?$")
# instructions whose effect does not depend on EXEC (or that have none): allowed in front of the restore
PROLOGUE_OK = re.compile(r"^\s*(v_writelane_b32|v_readlane_b32|v_readfirstlane_b32|s_\w+)\b")
# the prologue of a join block ends where EXEC is written by anything but the restore (`s_and_saveexec` opens a nested
# region: an `s_or_b64 exec` behind it closes THAT region, and what stands in front of it ran under the block's full mask) or
# where control leaves the block
# (a CONDITIONAL scalar branch does not end it: the fall-through still reaches the restore with the same suspect copies in
# front of it -- round 5, after review; only an unconditional transfer does)
PROLOGUE_END = re.compile(r"^\s*(s_\w+_saveexec_b64\b|s_\w+\s+exec\b|s_branch\b|s_endpgm\b|s_setpc_b64\b)")
# everything else that executes per lane
EXEC_DEPENDENT = re.compile(r"^\s*(v_|ds_|global_|buffer_|scratch_|flat_|tbuffer_|image_)\w+")


def scan(path):
    with open(path) as fh:
        lines = fh.read().split("\n")
    join_targets = set()
    func = None
    for line in lines:
        m = LABEL.match(line)
        if m and not m.group(1).startswith(".L"):
            func = m.group(1)
        t = EXECZ_TARGET.match(line)
        if t:
            join_targets.add((func, t.group(1)))
    findings = []
    func, block, pending = None, None, []
    for ln, line in enumerate(lines, 1):
        m = LABEL.match(line)
        if m:
            name = m.group(1)
            if not name.startswith(".L"):
                func = name
            if name.startswith(".L") and block is not None and pending and (func, name) not in join_targets:
                continue                                   # fall-through of a conditional branch into the next label: same prologue
            block, pending = (name if (func, name) in join_targets else None), []
            continue
        if block is None:
            continue
        s = line.split(";", 1)[0]
        if not s.strip() or s.lstrip().startswith("."):
            continue
        if EXEC_RESTORE.match(s):
            for p in pending:
                findings.append((func, block) + p)
            block = None
            continue
        if PROLOGUE_END.match(s):
            block = None
            continue
        c = COPY.match(s)
        if c and not VECTOR_SRC.match(c.group(3)):        # a splat of an SGPR / constant: uniform rematerialisation, harmless
            continue
        if PROLOGUE_OK.match(s):
            continue
        if c or EXEC_DEPENDENT.match(s):
            pending.append((ln, s.strip()))
            continue
        block = None                                       # anything else (unknown mnemonic): stop, do not guess
    return findings


def main(argv):
    bad = 0
    for path in argv:
        for func, block, ln, ins in scan(path):
            print(f"{path}:{ln}: {func} {block}: EXEC-dependent copy `{ins}` in front of the join block's exec restore")
            bad += 1
    print(f"{bad} suspect instruction(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
