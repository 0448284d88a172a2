#!/usr/bin/env python3
"""Per-basic-block opcode histogram of one kernel in a hipcc -S file: where the hot loop's instructions go.
   tools/isa_blocks.py file.s <mangled-or-demangled substring> [min instructions per block]"""
import collections, re, subprocess, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 16
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    if flt not in dn and flt not in name:
        continue
    print("==", dn[:140])
    blocks, cur = [], ('entry', [])
    for ln in body.split('s_endpgm')[0].splitlines():
        t = ln.strip()
        if not t or (t.startswith((';', '.')) and not t.endswith(':')):
            continue
        if t.endswith(':') or re.match(r'^\.LBB\S+:', t):
            blocks.append(cur); cur = (t.split(':')[0], []); continue
        cur[1].append(t)
    blocks.append(cur)
    for lab, ins in blocks:
        if len(ins) < minn:
            continue
        c = collections.Counter(i.split()[0] for i in ins)
        v = sum(n for k, n in c.items() if k.startswith('v_'))
        br = [i for i in ins if i.startswith(('s_cbranch', 's_branch'))]
        print(f"{lab:12s} n={len(ins):4d} valu={v:4d} | " + ', '.join(f'{k}:{n}' for k, n in c.most_common(16)) + (f" | {br[-1]}" if br else ""))
