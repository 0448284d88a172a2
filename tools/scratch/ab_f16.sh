#!/bin/bash
# Run ON THE GPU BOX: the k19m launch and the whole step of library variants with --dtype $1 (f16 | f16x3 | f32).
DT=$1; shift
LIB=yolo-fastest-and-embedded-deployment_amd/libyolo_fastest_hip.so
cp $LIB /tmp/orig.so
for L in "$@"; do cp $L $LIB; python bench.py --dtype $DT --no-cpu-baseline --no-variants --no-configs --no-train --no-live-traffic --in-flight 1 --dump-ops /tmp/ops.json 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read()); o = json.load(open('/tmp/ops.json'))
print('$L', j['value'], 'k19m %.1f us' % (1e3 * o[2]['ms']), 'sum %.1f us' % (1e3 * sum(x['ms'] for x in o)))"; done
cp /tmp/orig.so $LIB
