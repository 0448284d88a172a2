#!/usr/bin/env python3
"""Per-kernel resource usage of one csrc file: VGPRs, SGPRs, scratch (spills), LDS, occupancy.
  python tools/kres.py yf_mres_kernels.hip [substring filter]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "yolo-fastest-and-embedded-deployment_amd", "csrc", sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
p = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-fast-math", "-ffp-contract=off",
                    "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/kres.o"] + sys.argv[3:], capture_output=True, text=True)
name, out = None, {}
for line in p.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = m.group(1); out[name] = {}
    for key, tag in (("VGPRs:", "vgpr"), ("AGPRs:", "agpr"), ("ScratchSize", "scratch"), ("Occupancy", "occ"), ("SGPRs:", "sgpr"), ("LDS Size", "lds")):
        m = re.search(key + r"\D*(\d+)", line)
        if m and name:
            out[name][tag] = int(m.group(1))
try:
    names = subprocess.run(["c++filt"], input="\n".join(out), capture_output=True, text=True).stdout.splitlines()
except FileNotFoundError:
    names = list(out)
for n, d in zip(out, names):
    d = d.replace("void yf::", "").split("(yf::")[0]
    if flt in d:
        v = out[n]
        print(f"{d[:84]:84s} vgpr {v.get('vgpr'):4d} agpr {v.get('agpr', 0):3d} sgpr {v.get('sgpr'):3d} scratch {v.get('scratch'):4d} occ {v.get('occ')}")
