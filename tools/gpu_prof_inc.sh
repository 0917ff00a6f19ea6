# PMC passes over the incremental step kernel (tools/inc_bench.py <d>): outputs under
# gpurun_out/prof_inc/.   usage: bash tools/gpu_prof_inc.sh [d]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
D=${1:-30}
OUT=gpurun_out/prof_inc; rm -rf $OUT; mkdir -p $OUT
CMD="python tools/inc_bench.py $D"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU -d $OUT/pmc1 -o p -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE -d $OUT/pmc2 -o p -- $CMD > $OUT/pmc2.log 2>&1
python tools/pmc_summary.py $OUT step_ | tee $OUT/summary.txt
