#!/bin/bash
# Interleaved A/B of the number of batches in flight (and lanes per batch) on the headline workload.
# usage (on the GPU box, from the repo root): tools/inflight_ab.sh [rounds]
R=${1:-2}
F="--no-cpu-baseline --no-variants --no-live-traffic --no-train --no-configs --no-extras --steps 40 --warmup 10"
for r in $(seq $R); do
  for cfg in "2 1" "3 1" "4 1" "2 2" "3 2"; do
    set -- $cfg
    v=$(python bench.py $F --in-flight $1 --lanes $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
    echo "round $r in_flight=$1 lanes=$2 : $v"
  done
done
