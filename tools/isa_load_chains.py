#!/usr/bin/env python3
"""Finds chains of DEPENDENT-looking memory round trips in straight-line gfx950 ISA: global load(s) -> `s_waitcnt vmcnt(0)` -> a few
instructions -> global load(s) -> `s_waitcnt vmcnt(0)` ..., i.e. places where the compiler waits for one batch of loads before it issues
the next although the source does not need it to (loads under per-element bounds branches, a look-up behind each load: the u8 stem of
round 4 had nine in a row).  Reports, per kernel, every run of >= 3 (load, full wait) alternations with no loop back-edge in between.
   python tools/isa_load_chains.py [file.hip ...]   (default: the inference kernels of csrc/)"""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "yolo-fastest-and-embedded-deployment_amd", "csrc")
files = sys.argv[1:] or [f for f in sorted(glob.glob(os.path.join(CS, "*.hip"))) if "train" not in f and "engine" not in f]
for f in files:
    f = f if os.path.isabs(f) else os.path.join(CS, f)
    asm = "/tmp/isa_chains.s"
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-fast-math", "-ffp-contract=off", "-S",
                        "--cuda-device-only", f, "-o", asm], capture_output=True, text=True)
    if r.returncode:
        print(f, "does not compile"); continue
    text = open(asm).read()
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel', text, re.S | re.M):
        code = [l.strip() for l in m.group(2).split('s_endpgm')[0].splitlines() if l.strip() and not l.strip().startswith(';')]
        events = []   # ('L', i) load, ('W', i) full vm wait, ('B', i) backward branch / barrier
        labels = {}
        for i, l in enumerate(code):
            if l.endswith(':'): labels[l[:-1]] = i
        for i, l in enumerate(code):
            if l.startswith(("global_load", "buffer_load")): events.append(('L', i))
            elif l.startswith("s_waitcnt") and "vmcnt(0)" in l: events.append(('W', i))
            elif l.startswith(("s_cbranch", "s_branch")):
                tgt = l.split()[-1]
                if tgt in labels and labels[tgt] <= i: events.append(('B', i))
        # collapse into alternations
        runs, cur, last = [], 0, None
        start = None
        for k, i in events:
            if k == 'B':
                if cur >= 3: runs.append((cur, start, i))
                cur, last, start = 0, None, None
            elif k == 'L':
                if last != 'L':
                    if start is None: start = i
                last = 'L'
            elif k == 'W' and last == 'L':
                cur += 1; last = 'W'
        if cur >= 3: runs.append((cur, start, len(code)))
        if runs:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            print(f"{os.path.basename(f)}: {name[10:120]}: " + ", ".join(f"{n} load->wait rounds in lines {a}..{b}" for n, a, b in runs))
