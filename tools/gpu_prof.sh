set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
CMD="python tools/quick_engine_bench.py 30 65536 64 300"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof/trace -o t -- $CMD > gpurun_out/prof/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU -d gpurun_out/prof/pmc1 -o p -- $CMD > gpurun_out/prof/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVES -d gpurun_out/prof/pmc2 -o p -- $CMD > gpurun_out/prof/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d gpurun_out/prof/pmc3 -o p -- $CMD > gpurun_out/prof/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_COUNT -d gpurun_out/prof/pmc4 -o p -- $CMD > gpurun_out/prof/pmc4.log 2>&1
grep -i "SMEM\|SCA\|K_\|DCACHE\|IFETCH" gpurun_out/prof/../prof/trace.log | head -3
rocprofv3 -L 2>/dev/null | grep -oE "SQC?_[A-Z_0-9]+" | sort -u | tr '\n' ' ' > gpurun_out/prof/counter_names.txt
