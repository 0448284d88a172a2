#!/bin/bash
# Run HERE after `gpurun -- 'bash tools/refresh_profiles.sh r03'` has merged gpurun_out/: turns the raw rocprofv3 output into the tracked
# summaries under profiles/ (kernel stats per run, PMC rows of the yf:: kernels, pmc_traffic.json stamped with the source hash).
set -e
cd "$(dirname "$0")/.."
TAG=${1:-r06}
python tools/pmc_traffic.py gpurun_out/ops.json gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv profiles/pmc_traffic.json gpurun_out/${TAG}_source_hash.txt 256 256 f32 | head -1
for t in default lanes1 f16 f16x3 f16x3_512; do
  f=$(ls gpurun_out/prof_$t/*_kernel_stats.csv)
  case $t in f16) n=${TAG}_f16_640x512;; f16x3_512) n=${TAG}_f16x3_640x512;; *) n=${TAG}_$t;; esac
  cp $f profiles/${n}_kernel_stats.csv
done
for t in train train256; do
  f=$(ls gpurun_out/prof_$t/*_kernel_stats.csv)
  case $t in train) n=${TAG}_train_step_batch16;; *) n=${TAG}_train_step_batch256;; esac
  cp $f profiles/${n}_kernel_stats.csv
  { echo "# tools/train_bench.py UNDER rocprofv3 --kernel-trace (tracing adds host time per launch; unprofiled numbers: DESIGN.md section 4)"; grep "^GPU" gpurun_out/prof_$t.log; } > profiles/${n}_bench_line.txt || true
done
python - <<PY
import csv
for src, dst in (('gpurun_out/pmc_fetch/f_counter_collection.csv', 'profiles/${TAG}_pmc_fetch_size.csv'), ('gpurun_out/pmc_write/w_counter_collection.csv', 'profiles/${TAG}_pmc_write_size.csv')):
    rows = list(csv.DictReader(open(src)))
    keep = [r for r in rows if 'yf::' in r['Kernel_Name']]
    cols = ['Dispatch_Id', 'Kernel_Name', 'Grid_Size', 'Workgroup_Size', 'LDS_Block_Size', 'VGPR_Count', 'SGPR_Count', 'Counter_Name', 'Counter_Value']
    with open(dst, 'w', newline='') as fh:
        w = csv.DictWriter(fh, fieldnames=cols); w.writeheader()
        for r in keep:
            r = dict(r); r['Kernel_Name'] = r['Kernel_Name'].replace('void yf::', '').split('(yf::')[0][:100]
            w.writerow({c: r[c] for c in cols})
PY
# overlap of the headline loop (two batches in flight): which kernels run alone, which beside another batch's
if ls gpurun_out/prof_headline/*_kernel_trace.csv >/dev/null 2>&1; then
  { echo "# tools/trace_overlap.py over the kernel trace of \`bench.py --headline-only --regions 1\` (the headline loop alone, two batches in flight); build $(cat gpurun_out/${TAG}_source_hash.txt)";
    python tools/trace_overlap.py $(ls gpurun_out/prof_headline/*_kernel_trace.csv | head -1) --steps 20 --pairs k19r; } > profiles/${TAG}_overlap_headline.txt
fi
# what binds BASELINE configs[2] per launch (tools/pmc_configs2.sh summarises on the box)
for DT in f16x3 f16 f32; do
  for ext in txt json; do [ -f gpurun_out/${TAG}_configs2_binding_$DT.$ext ] && cp gpurun_out/${TAG}_configs2_binding_$DT.$ext profiles/; done
  [ -f gpurun_out/${TAG}_c2_${DT}_640x512_lanes1_kernel_stats.csv ] && cp gpurun_out/${TAG}_c2_${DT}_640x512_lanes1_kernel_stats.csv profiles/
done
[ -f gpurun_out/${TAG}_bench_line.json ] && cp gpurun_out/${TAG}_bench_line.json profiles/
echo "profiled build: $(cat gpurun_out/${TAG}_source_hash.txt)   this tree: $(python -c 'import bench; print(bench.source_hash())')"
