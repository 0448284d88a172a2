#!/usr/bin/env python3
"""Per-kernel instruction mix from a hipcc -S file: total VALU, FMAs, SGPR-spill traffic (v_readlane / v_writelane), LDS, SMEM.
   tools/isa_mix.py file.s [substring filter]"""
import collections, re, subprocess, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r'^void yf::', '', dn); dn = re.sub(r'\(.*\)$', '', dn)
    if flt not in dn:
        continue
    c = collections.Counter(ln.split()[0] for ln in body.split('s_endpgm')[0].splitlines()
                            if ln.strip() and not ln.strip().startswith((';', '.')) and not ln.strip().endswith(':'))
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    fma = sum(v for k, v in c.items() if k.startswith(('v_fma', 'v_pk_fma')))
    lanes = c['v_readlane_b32'] + c['v_writelane_b32']
    print(f"{dn[:96]:96s} valu={valu:5d} fma={fma:4d} mfma={sum(v for k, v in c.items() if k.startswith('v_mfma')):4d} "
          f"sgpr-spill={lanes:4d} ds={sum(v for k, v in c.items() if k.startswith('ds_')):3d} smem={sum(v for k, v in c.items() if k.startswith('s_load')):3d}")
