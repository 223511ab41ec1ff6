#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
V=build/variants
timeout -k 10 400 python tools/kbench.py gemm --rounds 5 --tiles 1 --lib diffusionrenderer-comfyui_amd/libdrn.so --lib $V/libdrn_g1.so --lib $V/libdrn_g2.so --lib $V/libdrn_g3.so --lib $V/libdrn_g4.so --lib $V/libdrn_g7.so --lib $V/libdrn_g8.so 2>&1 | grep -v amdgpu.ids > gpurun_out/s7_gemm_abl.log || exit 3
cat gpurun_out/s7_gemm_abl.log
