"""Find register spills that the compiler placed where EXEC cannot cover the lanes they serve.

usage: python tools/check_spills.py listing.s [kernel_name_substring]

hipcc lowers divergent loops and branches to EXEC-mask updates; a basic block that rejoins
lanes begins (after a few scalar moves) with `s_or_b64 exec, exec, <saved mask>`.  A vector
register spill (`scratch_store ... Folded Spill`) that the register allocator puts into such a
block AHEAD of that instruction runs under the narrower mask: in the exit block of a loop that
is left lane by lane, under EXEC == 0, so nothing is stored, the register is then reused, and the
reload returns whatever the scratch slot held before.  That placement lost a wave-uniform queue
position in trace_kernel's general variant (wrong images, no fault).  The check is static: it
prints every such store and exits non-zero if there is one.
"""
import re
import sys

text = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
bad = 0
for f in re.split(r"\n(?=_Z[^\n]*:\s*;?\s*@)", text):
    name = f.split(":")[0]
    if want not in name or not name.startswith("_Z"):
        continue
    block, pending = "entry", []
    spills = 0
    for line in f.split("\n"):
        lab = re.match(r"\.(LBB\d+_\d+):", line)
        if lab or line.startswith("; %bb."):
            block, pending = (lab.group(1) if lab else line.strip()), []
            continue
        s = line.strip()
        if "Folded Spill" in s and s.startswith("scratch_store"):
            spills += 1
            pending.append(s)
        elif re.match(r"s_or_b64 exec, exec,", s) and pending:
            for p in pending:
                print(f"{name[:90]}\n    block {block}: `{p}` runs before `{s}`")
                bad += 1
            pending = []
        elif s.startswith("s_cbranch") or s.startswith("s_branch"):
            pending = []
print(f"{bad} vector spill(s) placed ahead of an EXEC restore")
sys.exit(1 if bad else 0)
