#!/bin/bash
# Run ON THE GPU BOX: k19r_kernel's workgroup shapes (YF_K19R_NW: 8 | 12 | 16 = weights from LDS | 1012 = 12 waves, weights from LDS), interleaved:
# the launch's time (single lane).      tools/k19r_nw.sh [rounds] [nw ...]
R=${1:-2}; shift
NWS=${@:-12 16 1012 8}
F="--no-cpu-baseline --no-variants --no-configs --no-train --no-live-traffic --no-extras"
for r in $(seq $R); do for v in $NWS; do
  export YF_K19R_NW=$v
  python bench.py $F --in-flight 1 --lanes 1 --steps 20 --dump-ops gpurun_out/k19nw_$v.json > /dev/null 2>&1
  python - $v <<'PY'
import json, sys
o = json.load(open(f"gpurun_out/k19nw_{sys.argv[1]}.json"))
k = [x for x in o if "conv1_9" in x["name"]][0]
print(f"YF_K19R_NW={sys.argv[1]}: {k['name']} {k['ms'] * 1e3:.1f} us; launch sum {sum(x['ms'] for x in o) * 1e3:.1f} us")
PY
done; done
