#!/bin/bash
# quick GPU check after a GEMM change: kernel tests + the DiT goldens + the config-1 bench line with and without the new path
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_dit_gpu.py -x -q -m gpu -k "gemm or full28 or golden or batch" > gpurun_out/chk_tests.log 2>&1
rc=$?
tail -8 gpurun_out/chk_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for v in 0 1; do
  DRN_GEMM_TALL=$v timeout -k 10 250 python bench.py --config cfg1 --steps 16 --warmup 4 --no-cpu-baseline --no-tokenizer --no-cfg 2>/dev/null | cut -c1-200 || exit 1
done
