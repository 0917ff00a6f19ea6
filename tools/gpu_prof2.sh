cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
for GS in 64 256; do
CMD="python tools/quick_engine_bench.py 30 65536 $GS 300"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SMEM -d gpurun_out/prof/a$GS -o p -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_REQ SQC_TC_DATA_READ_REQ SQC_DCACHE_BUSY_CYCLES SQC_TC_STALL SQC_DCACHE_MISSES_DUPLICATE -d gpurun_out/prof/b$GS -o p -- $CMD > /dev/null 2>&1
done
