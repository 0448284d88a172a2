#!/usr/bin/env python3
"""Per-kernel ISA statistics from a hipcc -S file: VGPR/SGPR, scratch, FMA / s_load / global_load counts."""
import re, subprocess, sys
s = open(sys.argv[1]).read()
# kernels: label line "name:" ... ".end_amdhsa_kernel"
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    code = body.split('s_endpgm')[0]
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r'^void yf::', '', dn); dn = re.sub(r'\(.*\)$', '', dn)
    g = lambda k: (re.search(r'\.amdhsa_' + k + r' (\d+)', body) or [0, '?'])[1]
    c = lambda pat: len(re.findall(pat, code))
    print(f"{dn[:70]:70s} vgpr={g('next_free_vgpr'):>4} sgpr={g('next_free_sgpr'):>4} scratch={g('private_segment_fixed_size'):>4} "
          f"fma={c(r'v_fma_f32|v_fmac_f32|v_pk_fma_f32'):5d} s_load={c(r's_load_dword'):4d} g_load={c(r'global_load'):3d} "
          f"ds={c(r'ds_read|ds_write'):3d} lines={len(code.splitlines())}")
