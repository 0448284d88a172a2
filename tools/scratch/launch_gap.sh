F="--no-cpu-baseline --no-variants --no-configs --no-train --no-live-traffic --no-extras --in-flight 1 --lanes 1 --steps 10 --warmup 3"
for nw in 16 12 8; do for rep in 1 4 16; do
  YF_K19R_NW=$nw python bench.py $F --launch-repeats $rep --dump-ops gpurun_out/gap.json > /dev/null 2>&1
  python - $nw $rep <<'PY'
import json,sys
o=json.load(open("gpurun_out/gap.json")); k=[x for x in o if "conv1_9" in x["name"]][0]; r=[x for x in o if x["name"].startswith("res3_3")][0]; st=o[0]
print(f"NW={sys.argv[1]} repeats={sys.argv[2]}: k19 {k['ms']*1e3:.1f} us  res3_3 {r['ms']*1e3:.1f}  stem {st['ms']*1e3:.1f}  sum {sum(x['ms'] for x in o)*1e3:.1f}")
PY
done; done
