set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/final
( while sleep 45; do echo "[tick] $(date +%T)"; done ) &
TICK=$!
trap "kill $TICK" EXIT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/tok -o tok -- python3 tools/tokbench.py > gpurun_out/final/tok.out 2> gpurun_out/final/tok.err || { tail -5 gpurun_out/final/tok.err; exit 3; }
grep round gpurun_out/final/tok.out
find gpurun_out/final/tok -name "*kernel_trace.csv" -delete
timeout -k 10 500 python bench.py 2>gpurun_out/final/bench_default.err | tail -1 > gpurun_out/final/bench_default.json || exit 4
cut -c1-200 gpurun_out/final/bench_default.json
timeout -k 10 300 python bench.py --config cfg1 2>gpurun_out/final/bench_cfg1.err | tail -1 > gpurun_out/final/bench_cfg1.json || exit 5
cut -c1-200 gpurun_out/final/bench_cfg1.json
