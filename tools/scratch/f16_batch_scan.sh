#!/bin/bash
# ON THE GPU BOX: per-launch times of the fp16-storage pass at 640x512 for several batch sizes (fixed cost vs per-frame cost of a launch)
mkdir -p gpurun_out/bscan
for b in 32 64 128 256; do
  python bench.py --in-flight 1 --lanes 1 --no-cpu-baseline --no-variants --no-configs --no-train --no-live-traffic --no-extras --res 512 --batch $b --dtype ${DT:-f16} --steps 20 --dump-ops gpurun_out/bscan/ops_$b.json > /dev/null 2>&1
done
python - "$1" <<'PY'
import json, sys
flt = sys.argv[1] if len(sys.argv) > 1 else ""
bs = (32, 64, 128, 256)
runs = {b: json.load(open(f"gpurun_out/bscan/ops_{b}.json")) for b in bs}
print("%-46s" % "launch" + " ".join("%8d" % b for b in bs))
for i, o in enumerate(runs[128]):
    if flt and flt not in o["name"]:
        continue
    print("%-46s" % o["name"][:46] + " ".join("%8.1f" % (runs[b][i]["ms"] * 1e3) if i < len(runs[b]) else "       -" for b in bs))
print("%-46s" % "total" + " ".join("%8.1f" % (sum(x["ms"] for x in runs[b]) * 1e3) for b in bs))
PY
