#!/usr/bin/env python3
"""What binds each launch of a single-lane forward pass -- from the five passes of tools/pmc_configs2.sh for one dtype.

  tools/c2_report.py gpurun_out/r04_c2_<dtype> [out.json]

Per launch (matched to the plan's ops by dispatch order): duration (kernel trace of the counter-free --stats run), wave time split
parked (s_waitcnt / barrier) / issue-stalled / issuing, the time the VALU and the matrix pipe would need alone
(SQ_INSTS_VALU x 4 cycles, SQ_VALU_MFMA_BUSY_CYCLES per SIMD, 1024 SIMDs at 2.4 GHz), LDS instruction issue share, HBM bytes
(FETCH_SIZE x 2048 + WRITE_SIZE x 1024, the guide's gfx950 corrections) against 8 TB/s.  `binding` names the largest of
  mfma  = MFMA busy time / duration        valu = VALU issue time / duration       hbm = HBM time at 8 TB/s / duration
  lds   = LDS-instruction active share     latency = parked share (nothing issuing: memory / barrier waits)
so that bench.py's launch_roofline() can report a counter-backed bound instead of a mechanical one."""
import csv, json, sys, collections

pre = sys.argv[1]
ops = json.load(open(pre + "_ops.json"))
n = len(ops)
SIMDS, GHZ = 1024, 2.4


def rows(path):
    # (kernels instantiated on _Float16 come out mangled -- the demangler does not know DF16_)
    return [r for r in csv.DictReader(open(path)) if ("yf::" in r["Kernel_Name"] or r["Kernel_Name"].startswith("_ZN2yf"))
            and "post_kernel" not in r["Kernel_Name"] and "nms_sorted" not in r["Kernel_Name"] and "spin" not in r["Kernel_Name"]]


def counters(path):
    by = collections.defaultdict(dict)
    names = {}
    for r in rows(path):
        by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
        names[int(r["Dispatch_Id"])] = r["Kernel_Name"]
    ids = sorted(by)
    assert len(ids) % n == 0 and ids, (len(ids), n, path)
    passes = len(ids) // n
    out = []
    for i in range(n):
        acc = collections.defaultdict(float)
        for p in range(passes):
            for k, v in by[ids[p * n + i]].items():
                acc[k] += v / passes
        acc["_kernel"] = names[ids[i]]
        out.append(acc)
    return out


def durations(path):
    rs = rows(path)
    rs.sort(key=lambda r: int(r["Dispatch_Id"]))
    assert len(rs) % n == 0 and rs, (len(rs), n)
    passes = len(rs) // n
    skip = passes // 4          # the first passes are warm-up
    d = [[] for _ in range(n)]
    for p in range(skip, passes):
        for i in range(n):
            r = rs[p * n + i]
            d[i].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return [sorted(x)[len(x) // 2] / 1e3 for x in d]


wait = counters(pre + "_wait/p_counter_collection.csv")
issue = counters(pre + "_issue/p_counter_collection.csv")
fetch = counters(pre + "_fetch/p_counter_collection.csv")
write = counters(pre + "_write/p_counter_collection.csv")
dur = durations(pre + "_stats/p_kernel_trace.csv")
res = []
print(f"{'launch':40s} {'us':>7s} {'parked':>7s} {'stall':>6s} {'issue':>6s} | {'valu':>6s} {'mfma':>6s} {'lds':>5s} {'hbm':>5s} {'MB':>7s} -> binding")
tot = collections.defaultdict(float)
for i, o in enumerate(ops):
    w, s = wait[i], issue[i]
    wc = max(w["SQ_WAVE_CYCLES"], 1.0)
    us = dur[i]
    valu_us = s["SQ_INSTS_VALU"] * 4 / SIMDS / (GHZ * 1e3)
    mfma_us = s["SQ_VALU_MFMA_BUSY_CYCLES"] / SIMDS / (GHZ * 1e3)
    hbm = fetch[i]["FETCH_SIZE"] * 2048 + write[i]["WRITE_SIZE"] * 1024
    hbm_us = hbm / 8e12 * 1e6
    f = {"mfma": mfma_us / us, "valu": valu_us / us, "hbm": hbm_us / us, "lds": w["SQ_ACTIVE_INST_LDS"] / wc,
         "latency": w["SQ_WAIT_ANY"] / wc}
    # a pipe "binds" when the launch spends most of its time in it; otherwise the launch waits (latency)
    pipe = max(("mfma", "valu", "hbm", "lds"), key=lambda k: f[k])
    binding = pipe if f[pipe] >= 0.5 or f[pipe] >= f["latency"] else "latency"
    r = dict(name=o["name"], kernel=w["_kernel"].split("(")[0][:80], us=round(us, 1), parked=round(f["latency"], 3),
             stalled=round(w["SQ_WAIT_INST_ANY"] / wc, 3), issuing=round(w["SQ_ACTIVE_INST_ANY"] / wc, 3), valu_frac=round(f["valu"], 3),
             mfma_frac=round(f["mfma"], 3), lds_frac=round(f["lds"], 3), hbm_frac=round(f["hbm"], 3), hbm_bytes=round(hbm),
             insts_valu=round(s["SQ_INSTS_VALU"]), insts_mfma=round(s["SQ_INSTS_MFMA"]), insts_lds=round(s["SQ_INSTS_LDS"]),
             lds_bank_conflict_frac=round(s["SQ_LDS_BANK_CONFLICT"] / max(s["SQ_LDS_IDX_ACTIVE"], 1.0), 3), binding=binding)
    res.append(r)
    tot["us"] += us; tot["valu"] += valu_us; tot["mfma"] += mfma_us; tot["hbm"] += hbm_us; tot["bytes"] += hbm
    print(f"{o['name'][:40]:40s} {us:7.1f} {100*f['latency']:6.1f}% {100*r['stalled']:5.1f}% {100*r['issuing']:5.1f}% | "
          f"{100*f['valu']:5.1f}% {100*f['mfma']:5.1f}% {100*f['lds']:4.1f}% {100*f['hbm']:4.1f}% {hbm/1e6:7.1f} -> {binding}")
print(f"pass: {tot['us']:.0f} us of kernels; VALU issue {tot['valu']:.0f} us ({100*tot['valu']/tot['us']:.0f} %), MFMA busy {tot['mfma']:.0f} us "
      f"({100*tot['mfma']/tot['us']:.0f} %), HBM at 8 TB/s {tot['hbm']:.0f} us ({100*tot['hbm']/tot['us']:.0f} %), {tot['bytes']/1e9:.3f} GB")
if len(sys.argv) > 2:
    json.dump({"source": "tools/pmc_configs2.sh + tools/c2_report.py", "launches": res,
               "pass": {k: round(v, 1) for k, v in tot.items()}}, open(sys.argv[2], "w"), indent=1)
