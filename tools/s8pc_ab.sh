#!/bin/bash
# Run ON THE GPU BOX (round 6): the stride-8 residual / triple launches on the producer/consumer kernel against the
# two-barrier kernel: correctness of one variant on the golden heads, then per-launch times, three interleaved rounds.
LIB=yolo-fastest-and-embedded-deployment_amd/libyolo_fastest_hip.so
cp $LIB /tmp/keep.so
cp tools/variants/pcA.so $LIB
python -m pytest tests/test_gpu_parity.py -x -q -k "heads_match_reference_goldens or full_size_batch_properties" 2>&1 | tail -n 2
cp /tmp/keep.so $LIB
tools/ops_abn.sh 3 tools/variants/base.so tools/variants/pcA.so tools/variants/pcB.so tools/variants/pcC.so
