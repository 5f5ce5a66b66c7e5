"""Static instruction counts per loop of one kernel in a hipcc -S listing.

usage: python tools/isa_loops.py listing.s kernel_name_substring
Blocks are attributed to the innermost loop the compiler's comments name
("in Loop: Header=BBx_y Depth=d"); VALU / SALU / LDS / VMEM instructions are counted.
"""
import re
import sys
from collections import defaultdict

text = open(sys.argv[1]).read()
want = sys.argv[2]
funcs = re.split(r"\n(?=_Z[^\n]*:\s*;?\s*@)", text)
for f in funcs:
    name = f.split(":")[0]
    if want not in name:
        continue
    cur = "top"
    last_label = None
    counts = defaultdict(lambda: defaultdict(int))
    depth = {}
    for line in f.split("\n"):
        lab = re.match(r"\.L(BB\d+_\d+):", line)
        if lab:
            last_label = lab.group(1)
            cur = "top"
        if line.startswith("; %bb."):
            cur = "top"
        h = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", line)
        if h:
            cur = h.group(1)
            depth[cur] = int(h.group(2))
        hh = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", line)
        if hh and last_label:
            cur = last_label
            depth[cur] = int(hh.group(1))
        s = line.strip()
        if not line.startswith("\t") or not s or s.startswith(".") or s.startswith(";"):
            continue
        op = s.split()[0]
        kind = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") else
                "lds" if op.startswith("ds_") else "vmem")
        counts[cur][kind] += 1
        if kind == "valu":
            # issue cycles per wave64 as measured by tools/ubench/valu_rates.hip (rounded)
            if re.match(r"v_(rcp|rsq|sqrt)_f64", op):
                cyc = 18
            elif "_f64" in op or op.startswith("v_pk_") or re.match(r"v_mul_(lo|hi)_[ui]32|v_mad_[ui]64", op):
                cyc = 4
            else:
                cyc = 2
            counts[cur]["cycles"] += cyc
            if "_f64" in op:
                counts[cur]["f64"] += 1
    print(name[:70])
    for k, c in counts.items():
        print(f"  {k:12s} depth {depth.get(k, 0)}  valu {c['valu']:5d} salu {c['salu']:5d} lds {c['lds']:4d} vmem {c['vmem']:4d}  f64 {c['f64']:4d}  est. VALU cycles {c['cycles']:5d}")
