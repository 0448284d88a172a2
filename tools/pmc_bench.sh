#!/bin/bash
# Run ON THE GPU BOX: issue-side SQ counters of every launch of one single-lane forward pass (tools/pmc_show.py reads them).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace -d $R/gpurun_out/pmc_issue -o p --output-format csv -- python3 $R/bench.py --launch-repeats 1 --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-configs --no-train --no-extras --no-live-traffic --in-flight 1 --lanes 1 > $R/gpurun_out/pmc_issue.log 2>&1
tail -2 $R/gpurun_out/pmc_issue.log | cut -c1-300
