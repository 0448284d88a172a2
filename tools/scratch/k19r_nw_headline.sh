F="--no-cpu-baseline --no-variants --no-configs --no-train --no-live-traffic --no-extras --steps 60 --warmup 10"
for r in 1 2 3; do for nw in 16 1012 12; do
  echo -n "NW=$nw: "; YF_K19R_NW=$nw python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['one_batch_in_flight']['value'])"
done; done
