#!/usr/bin/env python3
"""Where the memory operations, waits and barriers of one kernel sit in hipcc's assembly (-save-temps .s file):
   python3 tools/asm_trace.py file.s substring-of-kernel-name [...]"""
import re, sys
s = open(sys.argv[1]).read()
for tag in sys.argv[2:]:
    m = re.search(r'^(_Z\S*' + re.escape(tag) + r'\S*):[^\n]*\n(.*?)\n\s*s_endpgm', s, re.S | re.M)
    if not m:
        print(tag, "not found"); continue
    b = m.group(2).split('\n')
    ev = []
    for i, l in enumerate(b):
        t = l.strip()
        if t.startswith(('buffer_load', 'global_load')): ev.append((i, 'L'))
        elif t.startswith(('buffer_store', 'global_store')): ev.append((i, 'S'))
        elif t.startswith('ds_read') or t.startswith('ds_load'): ev.append((i, 'r'))
        elif t.startswith('ds_write') or t.startswith('ds_store'): ev.append((i, 'w'))
        elif t.startswith('s_waitcnt') and 'vmcnt' in t: ev.append((i, 'W' + re.search(r'vmcnt\((\d+)\)', t).group(1)))
        elif t.startswith('s_barrier'): ev.append((i, 'B'))
        elif t.startswith('s_cbranch') or (t.endswith(':') and t.startswith('.LBB')): ev.append((i, t[:16]))
    out = []
    for i, e in ev:
        if out and out[-1][1] == e and e in 'LSrw': out[-1][2] += 1
        else: out.append([i, e, 1])
    nv = sum(1 for l in b if l.strip().startswith('v_'))
    print(f"== {tag}: {len(b)} lines, {nv} VALU")
    print(' '.join(f"{i}:{e}{'x' + str(n) if n > 1 else ''}" for i, e, n in out))
