#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_dit_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu -s -k "gemm or batch or inverse_node or determinism" > gpurun_out/s4_tests.log 2>&1
rc=$?
grep -E "passed|failed|FAILED|Error|assert|cfg3 full|clip" gpurun_out/s4_tests.log | tail -30
exit $rc
