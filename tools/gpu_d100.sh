# d = 100 (BASELINE config 4): bench line, kernel-trace stats and SQ counters of the MFMA kernel
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/d100; rm -rf $OUT; mkdir -p $OUT
timeout 900 python bench.py --dim 100 --steps 40 --warmup 4 --cpu-seconds 8 > $OUT/bench.json 2> $OUT/bench.err
CMD="python bench.py --dim 100 --no-cpu-baseline --steps 10 --warmup 2"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc_sq -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- $CMD > /dev/null 2>&1
python tools/pmc_summary.py $OUT step_mfma
tail -2 $OUT/bench.err
