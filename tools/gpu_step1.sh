#!/bin/bash
# round-2 GPU step 1: kernel parity tests, then interleaved A/B timing of the attention / GEMM variants
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -s > gpurun_out/s1_kernels.log 2>&1
rc=$?
tail -5 gpurun_out/s1_kernels.log
if [ $rc -gt 1 ]; then echo "kernel tests ended with rc=$rc: stopping"; exit $rc; fi
V=build/variants
timeout -k 10 300 python tools/kbench.py attn --rounds 7 --lib $V/libdrn_r1attn.so --lib diffusionrenderer-comfyui_amd/libdrn.so --lib $V/libdrn_prio.so --lib $V/libdrn_epi8.so > gpurun_out/s1_kb_attn.log 2>&1 || exit 3
cat gpurun_out/s1_kb_attn.log
timeout -k 10 300 python tools/kbench.py gemm --rounds 7 --tiles 1 --lib $V/libdrn_r1g256.so --lib diffusionrenderer-comfyui_amd/libdrn.so > gpurun_out/s1_kb_gemm.log 2>&1 || exit 4
cat gpurun_out/s1_kb_gemm.log
exit $rc
