#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes (each in a run of its own; counter collection serialises the kernels, so the profiled command is
# the bench at 4 of the 28 blocks: per-launch traffic of a kernel does not depend on the block count)
set -o pipefail
mkdir -p gpurun_out
P=gpurun_out/pmc
rm -rf $P; mkdir -p $P
export TMPDIR=/tmp
( while sleep 45; do echo "[tick] $(date +%T)"; done ) &
TICK=$!
trap "kill $TICK" EXIT
CMD="python3 bench.py --blocks 4 --steps 1 --warmup 1 --no-cpu-baseline --no-cfg"
echo "--- FETCH_SIZE pass"
timeout -k 10 420 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/f -o f -- $CMD > $P/f.json 2> $P/f.err || { tail -5 $P/f.err; exit 5; }
echo "--- WRITE_SIZE pass"
timeout -k 10 420 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $P/w -o w -- $CMD > $P/w.json 2> $P/w.err || { tail -5 $P/w.err; exit 6; }
F=$(find $P/f -name "*counter_collection.csv" | head -1); W=$(find $P/w -name "*counter_collection.csv" | head -1)
PMC_FORWARDS=2 PMC_BLOCKS=4 python3 tools/pmc_traffic.py $F $W $P/pmc_traffic.json
find $P -name "*kernel_trace.csv" -delete; find $P -name "*counter_collection.csv" -delete
du -sh $P
