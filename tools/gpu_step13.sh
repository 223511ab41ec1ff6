#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
rm -f gpurun_out/s13.log
for sg in 0 1 2 4 8; do
  echo "== DRN_GEMM_STAGGER=$sg" >> gpurun_out/s13.log
  DRN_GEMM_STAGGER=$sg timeout -k 10 300 python tools/kbench.py gemm --rounds 5 --tiles 3 2>&1 | grep -v amdgpu.ids >> gpurun_out/s13.log || exit 3
done
cat gpurun_out/s13.log
