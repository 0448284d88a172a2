#!/bin/bash
# Run ON THE GPU BOX: N default-bench runs of 100 steps each, prints the values and their median (A/B comparisons of a build).
N=${1:-5}
for i in $(seq $N); do python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])"; done | sort -n | awk '{v[NR]=$1; printf "%s ", $1} END {print " median", v[int((NR+1)/2)]}'
