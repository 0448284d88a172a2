#!/bin/bash
# Run ON THE GPU BOX: A/B of two builds of the library in ONE call (boxes differ by ~1 %, so never compare across calls).
#   tools/ab_bench.sh [runs] [lib_A.so lib_B.so ...]   -- each lib is copied over the in-tree .so in turn, interleaved rounds,
#   100-step default-bench runs; prints every value and the median per lib.  With no libs: the in-tree build only.
#   AB_ARGS: extra bench.py arguments (e.g. "--res 512 --batch 128 --dtype f16x3")
N=${1:-5}; shift
LIB=yolo-fastest-and-embedded-deployment_amd/libyolo_fastest_hip.so
run() { python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-variants --no-configs --no-train --no-live-traffic --no-extras $AB_ARGS 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])"; }
if [ $# -eq 0 ]; then for i in $(seq $N); do run; done | sort -n | awk '{v[NR]=$1; printf "%s ", $1} END {print " median", v[int((NR+1)/2)]}'; exit 0; fi
cp $LIB /tmp/orig.so
for i in $(seq $N); do for L in "$@"; do cp $L $LIB; echo "$L $(run)"; done; done > /tmp/ab.txt
cp /tmp/orig.so $LIB
for L in "$@"; do grep "^$L " /tmp/ab.txt | awk '{print $2}' | sort -n | awk -v l=$L '{v[NR]=$1; printf "%s ", $1} END {print " median", v[int((NR+1)/2)], l}'; done
