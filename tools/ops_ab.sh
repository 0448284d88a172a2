#!/bin/bash
# Run ON THE GPU BOX: per-launch times (single lane, HIP events) of two library builds side by side.
#   tools/ops_ab.sh a.so b.so
LIB=yolo-fastest-and-embedded-deployment_amd/libyolo_fastest_hip.so
cp $LIB /tmp/orig.so
for i in 1 2; do L=${!i}; cp $L $LIB; python bench.py --in-flight 1 --lanes 1 --no-cpu-baseline --no-variants --no-configs --no-train --no-live-traffic --steps 20 --dump-ops gpurun_out/ops_$i.json > /dev/null 2>&1; done
cp /tmp/orig.so $LIB
python - <<'PY'
import json
a=json.load(open('gpurun_out/ops_1.json')); b=json.load(open('gpurun_out/ops_2.json'))
for x,y in zip(a,b): print(f"{x['name'][:38]:38s} {x['ms']*1000:8.1f} {y['ms']*1000:8.1f}")
print(f"{'total':38s} {sum(x['ms'] for x in a)*1000:8.1f} {sum(y['ms'] for y in b)*1000:8.1f}")
PY
