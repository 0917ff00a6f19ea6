cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
for GS in 64; do
CMD="python tools/quick_engine_bench.py 30 65536 $GS 300"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SMEM -d gpurun_out/prof/a$GS -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS -d gpurun_out/prof/b$GS -o p -- $CMD > /dev/null 2>&1
done
