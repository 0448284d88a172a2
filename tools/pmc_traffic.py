#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over `bench.py --lanes 1`.

  tools/pmc_traffic.py <ops.json> <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [source_hash.txt] [res batch dtype]

The output is stamped with the hash of the kernel sources the profiled build was made from (bench.source_hash(), written on the
GPU box by tools/refresh_profiles.sh) and with the workload; bench.py reports `roofline.traffic` from it only when both match.

Units and corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB of 64-B fabric requests;
on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read (16 B/lane), so it is
doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.  Launches of one forward pass are matched to the
plan's ops by order (single lane => deterministic order)."""
import csv, json, os, subprocess, sys

ops = json.load(open(sys.argv[1]))
n = len(ops)


def per_op(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and "yf::" in r["Kernel_Name"]
            and "post_kernel" not in r["Kernel_Name"] and "nms_sorted" not in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    vals = [float(r["Counter_Value"]) for r in rows]
    names = [r["Kernel_Name"] for r in rows]
    passes = len(vals) // n
    assert passes >= 1 and len(vals) % n == 0, (len(vals), n)
    avg = [sum(vals[p * n + i] for p in range(passes)) / passes for i in range(n)]
    return avg, names[:n], passes


fetch, knames, p1 = per_op(sys.argv[2], "FETCH_SIZE")
write, _, p2 = per_op(sys.argv[3], "WRITE_SIZE")
src_hash = open(sys.argv[5]).read().strip() if len(sys.argv) > 5 else None
wl = {"res": int(sys.argv[6]), "batch": int(sys.argv[7]), "dtype": sys.argv[8]} if len(sys.argv) > 8 else {"res": 256, "batch": 256, "dtype": "f32"}
try:
    commit = subprocess.check_output(["git", "-C", os.path.dirname(os.path.abspath(__file__)), "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    commit = None
out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --lanes 1",
       "source_hash": src_hash, "workload": wl, "git_head_when_processed": commit,
       "corrections": "bytes = FETCH_SIZE*1024*2 (gfx950 half-count of wide streaming reads) + WRITE_SIZE*1024",
       "passes_averaged": [p1, p2], "kernels": {}, "forward_total_hbm_bytes": 0.0, "forward_total_algorithmic_bytes": 0.0}
for i, o in enumerate(ops):
    b = fetch[i] * 1024 * 2 + write[i] * 1024
    k = out["kernels"].setdefault(o["name"], {"launches": 0, "hbm_bytes_per_launch": 0.0, "fetch_bytes": 0.0, "write_bytes": 0.0,
                                              "algorithmic_bytes_per_launch": o["algorithmic_bytes"], "kernel": knames[i].split("(")[0][:90]})
    k["launches"] += 1
    k["hbm_bytes_per_launch"] += b; k["fetch_bytes"] += fetch[i] * 2048; k["write_bytes"] += write[i] * 1024
    out["forward_total_hbm_bytes"] += b
    out["forward_total_algorithmic_bytes"] += o["algorithmic_bytes"]
for k in out["kernels"].values():  # same-named launches do not occur; keep per-launch means anyway
    for f in ("hbm_bytes_per_launch", "fetch_bytes", "write_bytes"):
        k[f] = round(k[f] / k["launches"])
out["forward_total_hbm_bytes"] = round(out["forward_total_hbm_bytes"])
json.dump(out, open(sys.argv[4], "w"), indent=1)
print("forward: HBM bytes %.1f MB vs algorithmic %.1f MB" % (out["forward_total_hbm_bytes"] / 1e6, out["forward_total_algorithmic_bytes"] / 1e6))
for name, k in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:8]:
    print(f"{name[:50]:50s} hbm {k['hbm_bytes_per_launch']/1e6:9.1f} MB  algorithmic {k['algorithmic_bytes_per_launch']/1e6:9.1f} MB")
