set -o pipefail
export TMPDIR=/tmp
P=gpurun_out/sq; rm -rf $P; mkdir -p $P
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $P/a -o a -- python3 tools/kbench.py gemm --rounds 1 > $P/a.out 2> $P/a.err || { tail -5 $P/a.err; exit 3; }
python3 tools/pmc_sq.py $(find $P/a -name "*counter_collection.csv" | head -1) | tee $P/gemm_sq_a.txt
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_LDS --output-format csv -d $P/b -o b -- python3 tools/kbench.py gemm --rounds 1 > $P/b.out 2> $P/b.err || { tail -5 $P/b.err; exit 4; }
python3 tools/pmc_sq.py $(find $P/b -name "*counter_collection.csv" | head -1) | tee $P/gemm_sq_b.txt
find $P -name "*counter_collection.csv" -delete
