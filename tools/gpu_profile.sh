#!/bin/bash
# rocprofv3 evidence of the bench command -> gpurun_out/prof: kernel stats of the headline config (incl. the tokenizer leg and the
# measured pass) and of config 1, then FETCH_SIZE / WRITE_SIZE / MFMA-busy counters in passes of their own (counter collection
# serialises the kernels: 4 of the 28 blocks - per-launch figures of a kernel do not depend on the block count)
set -o pipefail
mkdir -p gpurun_out
P=gpurun_out/prof
rm -rf $P; mkdir -p $P
export TMPDIR=/tmp
( while sleep 45; do echo "[tick] $(date +%T)"; done ) &
TICK=$!
trap "kill $TICK" EXIT
echo "--- kernel stats, cfg3 (headline) incl. the tokenizer leg"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $P/ks3 -o ks3 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cfg --no-pass > $P/ks3.json 2> $P/ks3.err || { tail -5 $P/ks3.err; exit 3; }
tail -1 $P/ks3.json | cut -c1-300
echo "--- kernel stats, cfg1"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $P/ks1 -o ks1 -- python3 bench.py --config cfg1 --steps 8 --warmup 2 --no-cpu-baseline --no-tokenizer --no-cfg > $P/ks1.json 2> $P/ks1.err || { tail -5 $P/ks1.err; exit 4; }
tail -1 $P/ks1.json | cut -c1-300
CMD="python3 bench.py --blocks 4 --steps 1 --warmup 1 --no-cpu-baseline --no-cfg --no-pass"
echo "--- FETCH_SIZE pass"
timeout -k 10 420 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/f -o f -- $CMD > $P/f.json 2> $P/f.err || { tail -5 $P/f.err; exit 5; }
echo "--- WRITE_SIZE pass"
timeout -k 10 420 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $P/w -o w -- $CMD > $P/w.json 2> $P/w.err || { tail -5 $P/w.err; exit 6; }
F=$(find $P/f -name "*counter_collection.csv" | head -1); W=$(find $P/w -name "*counter_collection.csv" | head -1)
PMC_FORWARDS=3 PMC_BLOCKS=4 python3 tools/pmc_traffic.py $F $W $P/pmc_traffic.json
echo "--- MFMA-busy pass"
timeout -k 10 420 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $P/m -o m -- $CMD --no-tokenizer > $P/m.json 2> $P/m.err || { tail -5 $P/m.err; exit 7; }
MC=$(find $P/m -name "*counter_collection.csv" | head -1); MT=$(find $P/m -name "*kernel_trace.csv" | head -1)
python3 tools/pmc_mfma.py $MC $MT $P/pmc_mfma.json
# keep the summaries small enough to travel back: drop the per-dispatch traces, keep stats + folded counters
find $P -name "*kernel_trace.csv" -delete; find $P -name "*counter_collection.csv" -delete
find $P -name "*kernel_stats.csv" | head; du -sh $P
