#!/bin/bash
# Run ON THE GPU BOX: per-launch times (single lane, HIP events, mean of the passes of `bench.py --dump-ops`) of N library builds, ROUNDS
# interleaved rounds in one call.    tools/ops_abn.sh ROUNDS a.so b.so c.so ... [-- filter-substring]
# OPS_ARGS: extra bench.py arguments (e.g. "--res 512 --batch 128 --dtype f16x3")
R=$1; shift
LIBS=(); FILTER=""
while [ $# -gt 0 ]; do if [ "$1" = "--" ]; then FILTER=$2; break; fi; LIBS+=($1); shift; done
LIB=yolo-fastest-and-embedded-deployment_amd/libyolo_fastest_hip.so
cp $LIB /tmp/orig.so
for r in $(seq $R); do for i in "${!LIBS[@]}"; do
  cp ${LIBS[$i]} $LIB
  python bench.py --in-flight 1 --lanes 1 --no-cpu-baseline --no-variants --no-configs --no-train --no-live-traffic --no-extras $OPS_ARGS --steps 20 --dump-ops gpurun_out/ops_${i}_$r.json > /dev/null 2>&1
done; done
cp /tmp/orig.so $LIB
python - "$R" "$FILTER" "${LIBS[@]}" <<'PY'
import json, sys
R, flt, libs = int(sys.argv[1]), sys.argv[2], sys.argv[3:]
runs = [[json.load(open(f"gpurun_out/ops_{i}_{r}.json")) for r in range(1, R + 1)] for i in range(len(libs))]
print(" " * 40 + " ".join(f"{l.split('/')[-1][:14]:>14s}" for l in libs))
for k, o in enumerate(runs[0][0]):
    if flt and flt not in o["name"]:
        continue
    print(f"{o['name'][:40]:40s}" + " ".join("%14s" % "/".join("%.1f" % (rr[k]["ms"] * 1e3) for rr in runs[i]) for i in range(len(libs))))
print(f"{'total':40s}" + " ".join("%14s" % "/".join("%.1f" % (sum(x["ms"] for x in rr) * 1e3) for rr in runs[i]) for i in range(len(libs))))
PY
