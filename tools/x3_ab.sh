#!/bin/bash
# Run ON THE GPU BOX (round 6): per-launch times of the f16x3 engine at 640x512 batch 128 for
#   k19m_kernel<x3_t> (YF_K19X=0) | k19x_kernel forms (YF_K19X=43 / 42 / 83 / 123) | the hi/lo split with / without v_fma_mix (two library builds)
# ROUNDS interleaved rounds.   tools/x3_ab.sh ROUNDS
R=${1:-2}
LIB=yolo-fastest-and-embedded-deployment_amd/libyolo_fastest_hip.so
cp $LIB /tmp/keep.so
CFG=("x3nomix:0" "x3nomix:43" "x3mix:0" "x3mix:43" "x3mix4:43" "x3mix:123")
for r in $(seq $R); do for i in "${!CFG[@]}"; do
  lib=${CFG[$i]%%:*}; form=${CFG[$i]##*:}
  cp tools/variants/$lib.so $LIB
  YF_K19X=$form python bench.py --in-flight 1 --lanes 1 --no-cpu-baseline --no-variants --no-configs --no-train --no-live-traffic --no-extras --res 512 --batch 128 --dtype f16x3 --steps 20 --dump-ops gpurun_out/x3_${i}_$r.json > /dev/null 2>&1
done; done
cp /tmp/keep.so $LIB
python - "$R" "${CFG[@]}" <<'PY'
import json, sys
R, cfgs = int(sys.argv[1]), sys.argv[2:]
runs = [[json.load(open(f"gpurun_out/x3_{i}_{r}.json")) for r in range(1, R + 1)] for i in range(len(cfgs))]
print(" " * 40 + " ".join(f"{c:>14s}" for c in cfgs))
for k, o in enumerate(runs[0][0]):
    print(f"{o['name'][:40]:40s}" + " ".join("%14s" % "/".join("%.1f" % (rr[k]["ms"] * 1e3) for rr in runs[i]) for i in range(len(cfgs))))
print(f"{'total':40s}" + " ".join("%14s" % "/".join("%.1f" % (sum(x["ms"] for x in rr) * 1e3) for rr in runs[i]) for i in range(len(cfgs))))
PY
