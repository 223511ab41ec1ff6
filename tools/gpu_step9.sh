#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python tools/kbench.py gemm --rounds 9 --tiles 3 --lib diffusionrenderer-comfyui_amd/libdrn.so --lib build/variants/libdrn_splain.so 2>&1 | grep -v amdgpu.ids > gpurun_out/s9_gemm.log || exit 3
cat gpurun_out/s9_gemm.log
