#!/bin/bash
# the parity figures the GPU tests print (e_ref / e_hip / hip-vs-ref16 per golden, attention worst-ulp, g4 uint8 diffs, tokenizer
# vs its oracle), captured from ONE `pytest -m gpu -s` run -> gpurun_out/parity.txt (copied to profiles/rNN_parity.txt)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu -s -p no:cacheprovider > gpurun_out/parity_full.log 2>&1
rc=$?
{
  echo "# parity figures printed by \`python -m pytest tests -m gpu -s\` on one MI355X ($(date -u +%F)); rc=$rc"
  echo "# e_ref = rel-L2(reference bf16 CPU path, fp32 oracle); e_hip = rel-L2(HIP path, fp32 oracle); test bound: e_hip <= 1.5 e_ref + 1e-3"
  grep -E "e_ref=|attention .*rel-L2|uint8 mean|x0 rel-L2|g4:|identical|prefix|batched vs single|passed|failed" gpurun_out/parity_full.log | grep -v "^tests/" | cut -c1-260
} > gpurun_out/parity.txt
tail -5 gpurun_out/parity.txt
exit $rc
