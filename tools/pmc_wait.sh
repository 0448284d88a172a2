#!/bin/bash
# Run ON THE GPU BOX: where the waves of every launch spend their time (single-stream pass): parked at s_waitcnt / barriers
# (SQ_WAIT_ANY), issue-stalled (SQ_WAIT_INST_ANY), issuing (SQ_ACTIVE_INST_ANY); tools/wait_report.py reads it.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace -d $R/gpurun_out/pmc_wait -o p --output-format csv -- python3 $R/bench.py --launch-repeats 1 --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-configs --no-train --no-extras --no-live-traffic --in-flight 1 --lanes 1 > $R/gpurun_out/pmc_wait.log 2>&1
tail -1 $R/gpurun_out/pmc_wait.log | cut -c1-200
