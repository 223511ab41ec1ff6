#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for v in diag d32 d64 d16 d0p; do
  echo "=== $v" >> gpurun_out/s5_diag.log
  timeout -k 10 300 python tools/kbench.py attn --rounds 1 --diag build/variants/libdrn_$v.so 2>&1 | grep -v amdgpu.ids >> gpurun_out/s5_diag.log || exit 3
done
cat gpurun_out/s5_diag.log
