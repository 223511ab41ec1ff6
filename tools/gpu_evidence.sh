#!/bin/bash
# micro-benchmark evidence for profiles/: per-shape GEMM / attention rates, per-layer tokenizer convolutions, per-rank compute
set -o pipefail
mkdir -p gpurun_out/ev
timeout -k 10 300 python tools/kbench.py gemm attn --tiles=-1 --cold 4 2>&1 | grep -v amdgpu > gpurun_out/ev/kbench.txt || exit 1
timeout -k 10 300 python tools/libgemm_ref.py 2>&1 | grep -v amdgpu > gpurun_out/ev/libgemm.txt || true
timeout -k 10 300 python tools/convbench.py 2>&1 | grep -v amdgpu > gpurun_out/ev/convbench.txt || exit 2
for w in 2 4 8; do timeout -k 10 250 python tools/rankbench.py --world $w 2>&1 | grep -v amdgpu > gpurun_out/ev/rankbench_world$w.txt || exit 3; done
tail -n 5 gpurun_out/ev/*.txt
