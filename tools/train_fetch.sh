#!/bin/bash
# Run ON THE GPU BOX: HBM bytes fetched (PMC FETCH_SIZE) by the training kernels of one build, summed per kernel name.
#   tools/train_fetch.sh TAG [lib.so]     (lib.so is copied over the in-tree library for the run)
R=${GRAFT_REPO_ROOT:-/root/repo}
LIB=$R/yolo-fastest-and-embedded-deployment_amd/libyolo_fastest_hip.so
TAG=$1
if [ -n "$2" ]; then cp $LIB /tmp/orig_lib.so; cp $2 $LIB; fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/fetch_$TAG -o f --output-format csv -- python3 $R/tools/train_bench.py --batch 64 --steps 2 --warmup 1 > $R/gpurun_out/fetch_$TAG.log 2>&1
if [ -n "$2" ]; then cp /tmp/orig_lib.so $LIB; fi
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open('$R/gpurun_out/fetch_$TAG/f_counter_collection.csv')))
acc = collections.Counter(); n = collections.Counter()
for r in rows:
    if r['Counter_Name'] == 'FETCH_SIZE' and 'yf::' in r['Kernel_Name']:
        k = r['Kernel_Name'].replace('void yf::', '').split('(')[0][:48]
        acc[k] += float(r['Counter_Value']) * 1024 * 2; n[k] += 1
tot = sum(acc.values())
print('$TAG: total fetched %.1f MB over 3 iterations' % (tot / 1e6))
for k, v in acc.most_common(8):
    print('  %-50s %8.1f MB  (%d launches)' % (k, v / 1e6, n[k]))
PY
