#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > gpurun_out/s8_tests.log 2>&1
rc=$?
grep -E "passed|failed|FAILED|Error|assert" gpurun_out/s8_tests.log | tail -20
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 400 python tools/kbench.py gemm --rounds 7 --tiles 1,3 2>&1 | grep -v amdgpu.ids > gpurun_out/s8_gemm.log || exit 3
cat gpurun_out/s8_gemm.log
exit $rc
