#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python tools/kbench.py gemm --tiles 3,4 --cold 4 --lib build/variants/libdrn_base.so --lib build/variants/libdrn_nt.so > gpurun_out/s16_kbench.log 2>&1
rc=$?
cat gpurun_out/s16_kbench.log
exit $rc
