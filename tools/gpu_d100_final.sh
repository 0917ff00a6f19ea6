cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
prof() {
  OUT=$1; shift
  rm -rf $OUT; mkdir -p $OUT
  CMD="python bench.py --no-cpu-baseline --no-variants --steps 20 --warmup 4 $*"
  echo "$CMD" > $OUT/cmd.txt
  timeout 600 $CMD > $OUT/bench.json 2> $OUT/bench.err
  rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SMEM -d $OUT/pmc_sq -o p -- $CMD > /dev/null 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE -d $OUT/pmc_lds -o p -- $CMD > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- $CMD > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- $CMD > /dev/null 2>&1
}
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/final_gpu_tests.log
cat gpurun_out/final_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1; tail -2 gpurun_out/final_smoke.log
prof gpurun_out/final_d100 --dim 100 --steps 10 --warmup 2
tail -c 300 gpurun_out/final_d100/bench.json
INC_ONLY=1 timeout 300 python tools/inc_bench.py 52 56 64 80 100 112 128 2>&1 | grep "d=" > gpurun_out/sweep_2wave.log
