#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
rm -f gpurun_out/s11_rank.log
for w in 8 4 2; do
  timeout -k 10 300 python tools/rankbench.py --world $w --steps 3 2>&1 | grep -v "amdgpu.ids\|HSA version" >> gpurun_out/s11_rank.log || exit 3
done
timeout -k 10 300 python tools/rankbench.py --world 8 --exchange gather --steps 3 2>&1 | grep -v "amdgpu.ids\|HSA version" >> gpurun_out/s11_rank.log || exit 3
cat gpurun_out/s11_rank.log
