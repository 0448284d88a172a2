#!/bin/bash
# Run ON THE GPU BOX: the fp32 conv1_8+conv1_9+conv2_1 launch with k19r_kernel (default) and with the region-buffer k19m_kernel
# (YF_K19R=0), interleaved: per-launch time (single lane) and the headline.      tools/k19r_ab.sh [rounds]
R=${1:-2}
F="--no-cpu-baseline --no-variants --no-configs --no-train --no-live-traffic --no-extras"
for r in $(seq $R); do for v in 1 0; do
  export YF_K19R=$v
  python bench.py $F --in-flight 1 --lanes 1 --steps 20 --dump-ops gpurun_out/k19ab_$v.json > /dev/null 2>&1
  python - $v <<'PY'
import json, sys
o = json.load(open(f"gpurun_out/k19ab_{sys.argv[1]}.json"))
k = [x for x in o if "conv1_9" in x["name"]][0]
print(f"YF_K19R={sys.argv[1]}: {k['name']} {k['ms'] * 1e3:.1f} us; launch sum {sum(x['ms'] for x in o) * 1e3:.1f} us", end="; ")
PY
  python bench.py $F --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['value'], 'frames/s', d['ms_per_step'], 'ms/step')"
done; done
