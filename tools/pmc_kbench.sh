#!/bin/bash
# Run ON THE GPU BOX: SQ counters of one kbench section (tools/kb_plain <section>), two passes.  Output: gpurun_out/pmc_kb_{a,b}/
R=${GRAFT_REPO_ROOT:-/root/repo}
SEC=${1:-fbprod}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS --kernel-trace -d $R/gpurun_out/pmc_kb_a -o a --output-format csv -- $R/tools/kb_plain $SEC > $R/gpurun_out/pmc_kb_a.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY --kernel-trace -d $R/gpurun_out/pmc_kb_b -o b --output-format csv -- $R/tools/kb_plain $SEC > $R/gpurun_out/pmc_kb_b.log 2>&1
find $R/gpurun_out/pmc_kb_a $R/gpurun_out/pmc_kb_b -name "*.csv" | head
