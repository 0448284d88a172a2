for B in 256 64 32; do python bench.py --in-flight 1 --lanes 1 --no-cpu-baseline --no-variants --no-configs --no-train --no-live-traffic --no-extras --batch $B --steps 20 --dump-ops gpurun_out/ops_b$B.json > /dev/null 2>&1; done
for B in 128 32 16; do python bench.py --in-flight 1 --lanes 1 --no-cpu-baseline --no-variants --no-configs --no-train --no-live-traffic --no-extras --res 512 --batch $B --steps 20 --dump-ops gpurun_out/ops5_b$B.json > /dev/null 2>&1; done
python - <<'PY'
import json
for pre, Bs in (("ops_b", (256, 64, 32)), ("ops5_b", (128, 32, 16))):
    runs = {B: json.load(open(f"gpurun_out/{pre}{B}.json")) for B in Bs}
    print(pre, "us per frame at batch", Bs)
    for k, o in enumerate(runs[Bs[0]][:6]):
        print(f"{o['name'][:40]:40s}" + " ".join("%8.3f" % (runs[B][k]['ms'] * 1e3 / B) for B in Bs))
PY
