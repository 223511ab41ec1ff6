#!/bin/bash
# rocprofv3 evidence of the bench command (kernel stats; FETCH_SIZE / WRITE_SIZE in passes of their own) -> gpurun_out/prof_*
set -o pipefail
mkdir -p gpurun_out
P=gpurun_out/prof
rm -rf $P; mkdir -p $P
export TMPDIR=/tmp
echo "--- kernel stats, cfg3 (headline) incl. the tokenizer leg"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $P/ks3 -o ks3 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cfg > $P/ks3.json 2> $P/ks3.err || { tail -5 $P/ks3.err; exit 3; }
tail -1 $P/ks3.json | cut -c1-300
echo "--- kernel stats, cfg1"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $P/ks1 -o ks1 -- python3 bench.py --config cfg1 --steps 8 --warmup 2 --no-cpu-baseline --no-tokenizer --no-cfg > $P/ks1.json 2> $P/ks1.err || { tail -5 $P/ks1.err; exit 4; }
tail -1 $P/ks1.json | cut -c1-300
echo "--- FETCH_SIZE pass"
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/pmc_f -o f -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-cfg > $P/pmc_f.json 2> $P/pmc_f.err || { tail -5 $P/pmc_f.err; exit 5; }
echo "--- WRITE_SIZE pass"
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $P/pmc_w -o w -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-cfg > $P/pmc_w.json 2> $P/pmc_w.err || { tail -5 $P/pmc_w.err; exit 6; }
F=$(find $P/pmc_f -name "*counter_collection.csv" | head -1); W=$(find $P/pmc_w -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py $F $W $P/pmc_traffic.json
# keep the summaries small enough to travel back: drop the per-dispatch traces, keep stats + folded counters
find $P -name "*kernel_trace.csv" -delete; find $P -name "*counter_collection.csv" -delete
find $P -name "*kernel_stats.csv" | head; du -sh $P
