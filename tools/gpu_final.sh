# Round-end evidence run: GPU tests, smoke, the bench line, rocprofv3 kernel-trace stats and
# separate PMC passes of the SAME bench command.  Outputs under gpurun_out/final/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/final; rm -rf $OUT; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $OUT/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
CMD="python bench.py --no-cpu-baseline --steps 20 --warmup 4"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SMEM -d $OUT/pmc_sq -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE -d $OUT/pmc_lds -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- $CMD > /dev/null 2>&1
ls $OUT
