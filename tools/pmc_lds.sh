#!/bin/bash
# Run ON THE GPU BOX: LDS-side SQ counters of every launch of a single-lane forward pass (tools/lds_report.py reads them).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace -d $R/gpurun_out/pmc_lds -o p --output-format csv -- python3 $R/bench.py --launch-repeats 1 --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-configs --no-train --no-live-traffic --in-flight 1 --lanes 1 > $R/gpurun_out/pmc_lds.log 2>&1
tail -1 $R/gpurun_out/pmc_lds.log | cut -c1-200
