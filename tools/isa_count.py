#!/usr/bin/env python3
"""Static instruction counts per kernel of a hipcc -S listing (python tools/isa_count.py file.s [substring])."""
import re
import sys

text = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_ZN\S+):\s*;.*?\n(.*?)\n\.Lfunc_end", text, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if want not in name:
        continue
    ins = [l.strip() for l in body.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    v = sum(1 for l in ins if l.startswith("v_"))
    s = sum(1 for l in ins if l.startswith("s_"))
    g = sum(1 for l in ins if l.startswith(("global_", "buffer_", "flat_")))
    d = sum(1 for l in ins if l.startswith("ds_"))
    print(f"{name[:70]:70s} total {len(ins):5d}  valu {v:5d}  salu {s:5d}  vmem {g:4d}  lds {d:3d}")
