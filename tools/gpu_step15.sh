#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > gpurun_out/s15_tests.log 2>&1
rc=$?
tail -15 gpurun_out/s15_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/kbench.py gemm --tiles 3,4 --cold 4 > gpurun_out/s15_kbench.log 2>&1
rc=$?
cat gpurun_out/s15_kbench.log
exit $rc
