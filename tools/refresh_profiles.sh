#!/bin/bash
# Run ON THE GPU BOX (gpurun): regenerates the profile evidence under gpurun_out/ for the current build.
#   kernel-trace stats of the default bench run and of --in-flight 1 --lanes 1, PMC FETCH_SIZE / WRITE_SIZE passes (separate, as the
#   guide prescribes) with the per-launch op list.  Copy the summaries into profiles/ afterwards (tools/pmc_traffic.py).
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06}
python3 -c "import sys; sys.path.insert(0, '$R'); import bench; print(bench.source_hash())" > $R/gpurun_out/${TAG}_source_hash.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_default -o d --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-configs --no-train --no-extras --no-live-traffic > $R/gpurun_out/prof_default.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_lanes1 -o l --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-configs --no-train --no-extras --no-live-traffic --in-flight 1 --lanes 1 > $R/gpurun_out/prof_lanes1.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_f16 -o h --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-configs --no-train --no-extras --no-live-traffic --res 512 --batch 128 --dtype f16 > $R/gpurun_out/prof_f16.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_f16x3 -o x --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-configs --no-train --no-extras --no-live-traffic --dtype f16x3 > $R/gpurun_out/prof_f16x3.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_f16x3_512 -o y --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-configs --no-train --no-extras --no-live-traffic --res 512 --batch 128 --dtype f16x3 > $R/gpurun_out/prof_f16x3_512.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_fetch -o f --output-format csv -- python3 $R/bench.py --launch-repeats 1 --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-configs --no-train --no-extras --no-live-traffic --no-extras --in-flight 1 --lanes 1 --dump-ops $R/gpurun_out/ops.json > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_write -o w --output-format csv -- python3 $R/bench.py --launch-repeats 1 --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-configs --no-train --no-extras --no-live-traffic --in-flight 1 --lanes 1 > $R/gpurun_out/pmc_write.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_train -o t --output-format csv -- python3 $R/tools/train_bench.py --batch 16 --steps 20 > $R/gpurun_out/prof_train.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_train256 -o u --output-format csv -- python3 $R/tools/train_bench.py --batch 256 --steps 5 > $R/gpurun_out/prof_train256.log 2>&1
# the headline loop ALONE under the kernel trace (two batches in flight, the default scheduling): what tools/trace_overlap.py reads
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_headline -o hl --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --regions 1 --headline-only --no-live-traffic > $R/gpurun_out/prof_headline.log 2>&1
ls $R/gpurun_out/prof_default $R/gpurun_out/pmc_fetch $R/gpurun_out/prof_train $R/gpurun_out/prof_headline
