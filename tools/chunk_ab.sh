#!/bin/bash
# Run ON THE GPU BOX: BASELINE configs[2] (640x512 batch 128) with the batch cut into chunks that run one after the other through the layer
# chain (yf_set_chunk): do the stride-2 tensors of a chunk (stem -> res1_1 -> k19m: 2.6 MB per frame) stay in the 256 MB MALL?
R=${GRAFT_REPO_ROOT:-/root/repo}
F="--steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-configs --no-train --no-extras --no-live-traffic --res 512 --batch 128"
for round in 1 2; do
for dt in f16x3 f16 f32; do
for ch in 0 64 32; do
  python3 $R/bench.py $F --dtype $dt --chunk $ch 2>/dev/null | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$dt chunk $ch in_flight', j['config']['in_flight'], 'lanes', j['config']['lanes'], '->', j['value'], 'frames/s', j['ms_per_step'], 'ms/step; one at a time', j.get('one_batch_in_flight',{}).get('value'))"
done; done; done
