#!/bin/bash
# Build a VARIANT of the library for A/B runs on the GPU box (tools/ab_bench.sh, tools/ops_ab.sh):
#   tools/build_variant.sh NAME [extra hipcc flags, e.g. -DYF_RES2_TXB=20]   ->  tools/variants/NAME.so  (git-ignored, travels with gpurun)
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
SRC=$R/yolo-fastest-and-embedded-deployment_amd/csrc
OBJ=/tmp/yf_variant_$NAME
mkdir -p $OBJ $R/tools/variants
rm -f $OBJ/*.o $OBJ/*.fail
for f in $SRC/*.hip; do
  b=$(basename $f .hip)
  (/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=off -Wno-unused-function "$@" -c $f -o $OBJ/$b.o || touch $OBJ/$b.fail) &
done
wait
if ls $OBJ/*.fail >/dev/null 2>&1; then echo "COMPILE FAILED: $(ls $OBJ/*.fail)"; exit 1; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/variants/$NAME.so $OBJ/*.o
ls -la $R/tools/variants/$NAME.so
