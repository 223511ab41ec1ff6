#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -s > gpurun_out/s6_tests.log 2>&1
rc=$?
grep -E "passed|failed|FAILED|Error|assert" gpurun_out/s6_tests.log | tail -30
exit $rc
