#!/bin/bash
# Same-box A/B of the trainer's graph replay (GPU box): alternating runs of tools/train_bench.py with and without YF_TRAIN_GRAPH_OFF.
#   tools/graph_ab.sh [batch] [steps] [rounds]
B=${1:-16}; S=${2:-60}; R=${3:-3}
cat /proc/loadavg
for i in $(seq $R); do
  echo -n "off: "; YF_TRAIN_GRAPH_OFF=1 python tools/train_bench.py --batch $B --steps $S 2>&1 | tail -n 1
  echo -n "on : "; python tools/train_bench.py --batch $B --steps $S 2>&1 | tail -n 1
done
