#!/bin/bash
# Diagnosis of the capture crash: the engine with the branch issued while capturing (YF_CAPTURE_BRANCH=1: nested join as in eager mode;
# =2: branches join the origin stream directly), native backtrace on SIGSEGV (tools/libsegv_trace.so), through torch.cuda.graph
# (cap_try.py) and through the HIP runtime's capture calls alone (cap_try2.py).
cd "$(dirname "$0")/.."
for script in cap_try2.py cap_try.py; do
for mode in 1 2; do
  echo "======== $script YF_CAPTURE_BRANCH=$mode lanes 2 branches 1"
  YF_CAPTURE_BRANCH=$mode YF_SEGV_TRACE=1 timeout -k 5 120 python tools/$script 2 1 2>&1 | grep -v amdgpu.ids | head -80
  echo "rc=${PIPESTATUS[0]}"
done
done
exit 0
