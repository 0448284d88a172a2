#!/bin/bash
# Run ON THE GPU BOX: tools/cv_pre_bench.py for several builds of the library (tools/variants/cv_*.so)
LIB=yolo-fastest-and-embedded-deployment_amd/libyolo_fastest_hip.so
cp $LIB /tmp/keep.so
for L in tools/variants/cv_*.so; do cp $L $LIB; echo "== $L"; python tools/cv_pre_bench.py 2>/dev/null | head -n 5; done
cp /tmp/keep.so $LIB
