#!/usr/bin/env python3
"""Finds short rolled loops that load from global memory and wait for the load inside the same iteration (one memory round trip per
iteration, nothing in flight) in the gfx950 ISA of every kernel of a csrc file.  That is how `for (i ...) lds[i] = global[i];` compiles.
   python tools/isa_serial_loads.py [file.hip ...]   (default: every csrc/*.hip)"""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "yolo-fastest-and-embedded-deployment_amd", "csrc")
files = sys.argv[1:] or sorted(glob.glob(os.path.join(CS, "*.hip")))
for f in files:
    f = f if os.path.isabs(f) else os.path.join(CS, f)
    asm = "/tmp/isa_serial.s"
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-fast-math", "-ffp-contract=off", "-S",
                        "--cuda-device-only", f, "-o", asm], capture_output=True, text=True)
    if r.returncode:
        print(f, "does not compile:", r.stderr[-300:]); continue
    text = open(asm).read()
    kern = None
    lines = text.splitlines()
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kern = m.group(1)
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if not m:
            continue
        body = []
        for l2 in lines[i + 1:i + 40]:
            body.append(l2.strip())
            if re.match(r"s_cbranch\w+\s+" + re.escape(m.group(1)) + r"\b", l2.strip()):
                break
        else:
            continue
        loads = [b for b in body if b.startswith(("global_load", "buffer_load"))]
        waits = [b for b in body if b.startswith("s_waitcnt") and "vmcnt(0)" in b]
        if loads and waits:
            name = subprocess.run(["c++filt", kern or "?"], capture_output=True, text=True).stdout.strip()
            print(f"{os.path.basename(f)}: {name[:110]}: loop {m.group(1)} ({len(body)} instructions, {len(loads)} load(s), waits for them in the iteration)")
