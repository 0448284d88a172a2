#!/bin/bash
# ON THE GPU BOX: the fp16-storage pass at 640x512 batch 128, per-launch times, with and without an environment switch
#   tools/scratch/f16_env_ab.sh "VAR=1" [filter]
mkdir -p gpurun_out/envab
A="--in-flight 1 --lanes 1 --no-cpu-baseline --no-variants --no-configs --no-train --no-live-traffic --no-extras --res 512 --batch 128 --dtype ${DT:-f16} --steps 20"
for r in 1 2; do
  python bench.py $A --dump-ops gpurun_out/envab/ops_a_$r.json > /dev/null 2>&1
  env $1 python bench.py $A --dump-ops gpurun_out/envab/ops_b_$r.json > /dev/null 2>&1
done
python - "$2" <<'PY'
import json, sys
flt = sys.argv[1] if len(sys.argv) > 1 else ""
runs = {k: [json.load(open(f"gpurun_out/envab/ops_{k}_{r}.json")) for r in (1, 2)] for k in "ab"}
for i, o in enumerate(runs["a"][0]):
    if flt and flt not in o["name"]:
        continue
    print("%-46s %s   |   %s" % (o["name"][:46], " ".join("%6.1f" % (rr[i]["ms"] * 1e3) for rr in runs["a"]), " ".join("%6.1f" % (rr[i]["ms"] * 1e3) for rr in runs["b"] if i < len(rr))))
print("%-46s %s   |   %s" % ("total", " ".join("%6.1f" % (sum(x["ms"] for x in rr) * 1e3) for rr in runs["a"]), " ".join("%6.1f" % (sum(x["ms"] for x in rr) * 1e3) for rr in runs["b"])))
PY
