#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python tools/kbench.py gemm --rounds 5 --tiles 1,3 --cold 5 2>&1 | grep -v amdgpu.ids > gpurun_out/s10_gemm_cold.log || exit 3
cat gpurun_out/s10_gemm_cold.log
for m in 0 1; do
  DRN_GEMM_STREAM=$m timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-tokenizer --no-cfg 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('DRN_GEMM_STREAM=$m', r['value'], r['ms_per_step'], r['roofline']['per_kernel'])" | tee -a gpurun_out/s10_bench.log || exit 4
done
