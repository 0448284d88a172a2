#!/bin/bash
# Run ON THE GPU BOX: the fp32 headline with k19r_kernel (default) and with the region-buffer k19m_kernel (YF_K19R=0), ABBA-interleaved
# 100-step runs.      tools/k19r_ab.sh [rounds]
R=${1:-3}
F="--no-cpu-baseline --no-variants --no-configs --no-train --no-live-traffic --no-extras --steps 100 --warmup 10"
run() { YF_K19R=$1 python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('YF_K19R=$1', d['value'], 'frames/s', d['ms_per_step'], 'ms/step; one batch at a time', d['one_batch_in_flight']['value'], '; launch sum', d['forward_chain']['sum_of_launch_ms_single_stream'])"; }
for r in $(seq $R); do run 1; run 0; run 0; run 1; done
