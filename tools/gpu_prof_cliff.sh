# PMC passes over the step kernel of one tools/cliff_bench.py configuration: outputs under
# gpurun_out/prof_cliff/.   usage: bash tools/gpu_prof_cliff.sh d:K:n_periodic [more ...]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/prof_cliff; rm -rf $OUT; mkdir -p $OUT
CMD="python tools/cliff_bench.py $*"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU -d $OUT/pmc1 -o p -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $OUT/pmc2 -o p -- $CMD > $OUT/pmc2.log 2>&1
python tools/pmc_summary.py $OUT step_ | tee $OUT/summary.txt
rm -rf $OUT/pmc1 $OUT/pmc2
