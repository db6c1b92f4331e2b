#!/usr/bin/env python3
"""LDS opcode histogram per kernel of hipcc -S output: python3 tools/isa_lds.py file.s [...]   (ds_read2_b64 runs at half the
bytes per clock of ds_read_b64: the fused form is worth un-fusing in LDS-bound loops)."""
import re, sys
for f in sys.argv[1:]:
    txt = open(f).read()
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)s_endpgm', txt, re.S | re.M):
        name, body = m.group(1), m.group(2)
        ops = {}
        for o in re.findall(r'\b(ds_\w+)', body):
            ops[o] = ops.get(o, 0) + 1
        print(f.split('/')[-1], name[14:76])
        print("    " + "  ".join(f"{k} {v}" for k, v in sorted(ops.items())))
