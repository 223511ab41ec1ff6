#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
V=build/variants
timeout -k 10 300 python tools/kbench.py attn --rounds 7 --lib diffusionrenderer-comfyui_amd/libdrn.so --lib $V/libdrn_dmas1.so --lib $V/libdrn_dmas2.so --lib $V/libdrn_dmas4.so 2>&1 | grep -v amdgpu.ids > gpurun_out/s12.log || exit 3
timeout -k 10 300 python tools/kbench.py gemm --rounds 5 --tiles 3 --lib diffusionrenderer-comfyui_amd/libdrn.so --lib $V/libdrn_s8.so 2>&1 | grep -v amdgpu.ids >> gpurun_out/s12.log || exit 3
cat gpurun_out/s12.log
