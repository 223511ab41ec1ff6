#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -s -k "attention" > gpurun_out/s3_kernels.log 2>&1
rc=$?
grep -E "^attention|passed|failed|FAILED|Error" gpurun_out/s3_kernels.log | tail -40
if [ $rc -gt 1 ]; then echo "kernel tests ended with rc=$rc: stopping"; exit $rc; fi
V=build/variants
timeout -k 10 300 python tools/kbench.py attn --rounds 5 --lib $V/libdrn_r1attn.so --lib diffusionrenderer-comfyui_amd/libdrn.so --lib $V/libdrn_prio0.so --lib $V/libdrn_abl1.so --lib $V/libdrn_abl12.so --lib $V/libdrn_abl16.so > gpurun_out/s3_kb_attn.log 2>&1 || exit 3
cat gpurun_out/s3_kb_attn.log
exit $rc
