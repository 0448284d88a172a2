#!/bin/bash
# Run ON THE GPU BOX: what binds BASELINE configs[2] (640x512 batch 128 on the fp16 matrix pipe)?  Per dtype (f16x3, f16) one
# single-lane pass each with the wave-time counters (parked / issue-stalled / issuing), the issue-side counters (VALU and MFMA
# instructions, MFMA busy cycles, LDS), and the HBM byte counters (FETCH_SIZE, WRITE_SIZE in separate passes, as the guide prescribes).
# tools/c2_report.py turns them into profiles/<tag>_configs2_binding_<dtype>.txt (tools/profiles_collect.sh).
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
COMMON="--steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-configs --no-train --no-extras --no-live-traffic --in-flight 1 --lanes 1 --res 512 --batch 128"
for DT in f16x3 f16 f32; do
  O=$R/gpurun_out/${TAG}_c2_$DT
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace -d ${O}_wait -o p --output-format csv -- python3 $R/bench.py --launch-repeats 1 $COMMON --dtype $DT --dump-ops ${O}_ops.json > ${O}_wait.log 2>&1 || exit 1
  echo "$DT wait done"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d ${O}_issue -o p --output-format csv -- python3 $R/bench.py --launch-repeats 1 $COMMON --dtype $DT > ${O}_issue.log 2>&1 || exit 1
  echo "$DT issue done"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d ${O}_fetch -o p --output-format csv -- python3 $R/bench.py --launch-repeats 1 $COMMON --dtype $DT > ${O}_fetch.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d ${O}_write -o p --output-format csv -- python3 $R/bench.py --launch-repeats 1 $COMMON --dtype $DT > ${O}_write.log 2>&1 || exit 1
  echo "$DT traffic done"
  rocprofv3 --kernel-trace --stats -d ${O}_stats -o p --output-format csv -- python3 $R/bench.py --launch-repeats 1 --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-configs --no-train --no-extras --no-live-traffic --in-flight 1 --lanes 1 --res 512 --batch 128 --dtype $DT > ${O}_stats.log 2>&1 || exit 1
  tail -1 ${O}_stats.log | cut -c1-300
  # summarise ON THE BOX (the raw counter files of 15 passes exceed what gpurun copies back) and drop the raw files
  ( cd $R && { echo "# tools/c2_report.py over the five counter passes of tools/pmc_configs2.sh, 640x512 batch 128 $DT; build $(python3 -c "import sys; sys.path.insert(0, '$R'); import bench; print(bench.source_hash())")";
      python3 tools/c2_report.py gpurun_out/${TAG}_c2_$DT gpurun_out/${TAG}_configs2_binding_$DT.json; } > gpurun_out/${TAG}_configs2_binding_$DT.txt )
  cp ${O}_stats/p_kernel_stats.csv $R/gpurun_out/${TAG}_c2_${DT}_640x512_lanes1_kernel_stats.csv
  rm -rf ${O}_wait ${O}_issue ${O}_fetch ${O}_write ${O}_stats ${O}_*.log
done
