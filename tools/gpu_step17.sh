#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_vae_gpu.py -x -q -m gpu > gpurun_out/s17_tests.log 2>&1
rc=$?
tail -15 gpurun_out/s17_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/convbench.py > gpurun_out/s17_convbench.log 2>&1
rc=$?
head -24 gpurun_out/s17_convbench.log
exit $rc
