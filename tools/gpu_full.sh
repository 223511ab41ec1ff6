#!/bin/bash
# whole GPU suite + default bench line
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu -s > gpurun_out/full_tests.log 2>&1
rc=$?
grep -E "passed|failed|FAILED|Error|e_ref|g4:|cfg3|full28" gpurun_out/full_tests.log | tail -30
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 600 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
rc=$?
tail -3 gpurun_out/bench_full.err
cat gpurun_out/bench_full.json
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 400 python bench.py --config cfg1 > gpurun_out/bench_cfg1.json 2> gpurun_out/bench_cfg1.err
rc=$?
tail -3 gpurun_out/bench_cfg1.err
cat gpurun_out/bench_cfg1.json
exit $rc
