#!/bin/bash
# Run ON THE GPU BOX: A/B of library builds in PIPELINED mode (two batches in flight), interleaved rounds.
#   tools/ab_inflight.sh [rounds] lib_A.so lib_B.so ...
N=${1:-3}; shift
LIB=yolo-fastest-and-embedded-deployment_amd/libyolo_fastest_hip.so
run() { python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-variants --no-configs --no-train --no-live-traffic ${YF_BENCH_ARGS} 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['one_batch_in_flight']['value'] if 'one_batch_in_flight' in j else '')"; }
cp $LIB /tmp/orig.so
for i in $(seq $N); do for L in "$@"; do cp $L $LIB; echo "$L $(run)"; done; done | tee /tmp/abi.txt
cp /tmp/orig.so $LIB
