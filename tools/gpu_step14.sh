#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_parallel_gpu.py tests/test_vae_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu -s > gpurun_out/s14_tests.log 2>&1
rc=$?
grep -E "passed|failed|FAILED|Error" gpurun_out/s14_tests.log | tail -30
if [ $rc -ne 0 ]; then tail -40 gpurun_out/s14_tests.log; exit $rc; fi
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
