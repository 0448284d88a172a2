#!/bin/bash
# Run ON THE GPU BOX: A/B of ENVIRONMENT settings (developer switches read once per process) on the headline loop, interleaved rounds.
#   tools/env_ab.sh ROUNDS "A=1" "B=2 C=3" ...     ("-" = no setting)
N=${1:-3}; shift
for i in $(seq $N); do for E in "$@"; do
  if [ "$E" = "-" ]; then V=$(python bench.py --steps 100 --warmup 10 --headline-only --regions 1 $AB_ARGS 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
  else V=$(env $E python bench.py --steps 100 --warmup 10 --headline-only --regions 1 $AB_ARGS 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])"); fi
  echo "$E $V"
done; done > /tmp/envab.txt
for E in "$@"; do grep -F -- "$E " /tmp/envab.txt | awk '{print $NF}' | sort -n | awk -v l="$E" '{v[NR]=$1; printf "%s ", $1} END {print " median", v[int((NR+1)/2)], " [" l "]"}'; done
