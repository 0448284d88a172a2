#!/bin/bash
# ON THE GPU BOX: the fp16-storage pass at 640x512 batch 128 per form of k19h_kernel (YF_K19H_FORM) and with k19m_kernel<half_t> (YF_K19R=0)
mkdir -p gpurun_out/k19h
A="--in-flight 1 --lanes 1 --no-cpu-baseline --no-variants --no-configs --no-train --no-live-traffic --no-extras --res 512 --batch 128 --dtype f16 --steps 20"
for r in 1 2; do
for f in old 43 44 42 84; do
  if [ $f = old ]; then YF_K19R=0 python bench.py $A --dump-ops gpurun_out/k19h/ops_${f}_$r.json > /dev/null 2>&1
  else YF_K19H_FORM=$f python bench.py $A --dump-ops gpurun_out/k19h/ops_${f}_$r.json > /dev/null 2>&1; fi
done; done
python - <<'PY'
import json
for f in "old 43 44 42 84".split():
    out = []
    for r in (1, 2):
        ops = json.load(open(f"gpurun_out/k19h/ops_{f}_{r}.json"))
        k = [o for o in ops if "conv1_9" in o["name"] or "k19" in o["name"]]
        out.append("%.1f / %.1f" % (sum(o["ms"] for o in k) * 1e3, sum(o["ms"] for o in ops) * 1e3))
    print(f, [o["name"] for o in k][:1], out)
PY
