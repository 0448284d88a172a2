#!/bin/bash
# Run ON THE GPU BOX: f16x3 engines with the res2 pair on the split-operand MFMA block kernel (default) or on the fp32 VALU block kernel
# (v=1: the default since round 4; v=0: YF_RES2_X3=1 restores round 3's MFMA form), interleaved, at configs[2] (640x512 batch 128) and at the headline size.
R=${GRAFT_REPO_ROOT:-/root/repo}
F="--steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-configs --no-train --no-extras --no-live-traffic --dtype f16x3"
for round in 1 2 3; do
for v in 0 1; do
for wl in "--res 512 --batch 128" "--res 256 --batch 256"; do
  if [ $v = 1 ]; then unset YF_RES2_X3; else export YF_RES2_X3=1; fi
  python3 $R/bench.py $F $wl 2>/dev/null | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('res2_valu=$v $wl ->', j['value'], 'frames/s', j['ms_per_step'], 'ms/step; one at a time', j.get('one_batch_in_flight',{}).get('value'), 'launch sum', j['forward_chain']['sum_of_launch_ms_single_stream'])"
done; done; done
