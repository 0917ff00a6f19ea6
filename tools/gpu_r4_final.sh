# Round-4 evidence run: GPU tests, smoke, the full bench line, then rocprofv3 kernel-trace stats
# and separate PMC passes of the SAME bench command for the headline config (both evaluation
# modes), d = 100 and the plik-lite workload.  Outputs under gpurun_out/final{,_full,_d100,_pl}/;
# tools/collect_evidence.py turns them into profiles/.   usage: bash tools/gpu_r3_final.sh [quick]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
prof() {  # prof <outdir> <bench args...>
  OUT=$1; shift
  rm -rf $OUT; mkdir -p $OUT
  CMD="python bench.py --no-cpu-baseline --no-variants --steps 20 --warmup 4 --cross-check-seconds 0.25 $*"
  echo "$CMD" > $OUT/cmd.txt
  timeout 600 $CMD > $OUT/bench.json 2> $OUT/bench.err
  rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SMEM -d $OUT/pmc_sq -o p -- $CMD > /dev/null 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE -d $OUT/pmc_lds -o p -- $CMD > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- $CMD > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- $CMD > /dev/null 2>&1
  if echo "$*" | grep -q pliklite; then
    rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES -d $OUT/pmc_mfma -o p -- $CMD > /dev/null 2>&1
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum -d $OUT/pmc_l2 -o p -- $CMD > /dev/null 2>&1
  fi
  # only the result databases travel back (gpurun_out is capped at 64 MiB)
  find $OUT -type f ! -name "*_results.db" ! -name "*.json" ! -name "*.txt" ! -name "*.err" ! -name "*.log" -delete
}
mkdir -p gpurun_out/final
if [ "$1" != "quick" ]; then
  timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/final_gpu_tests.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1
fi
prof gpurun_out/final
prof gpurun_out/final_full --evaluation full
prof gpurun_out/final_d100 --dim 100 --steps 10 --warmup 2
prof gpurun_out/final_pl --workload pliklite --steps 8 --warmup 2
[ -f gpurun_out/final_gpu_tests.log ] && cp gpurun_out/final_gpu_tests.log gpurun_out/final/gpu_tests.log
# summaries on the box (the databases exceed what gpurun carries back), then drop the databases
export EVIDENCE_DST=$PWD/gpurun_out/evidence EVIDENCE_COMMIT=$(cat tools/.evidence_commit 2>/dev/null)
rm -rf $EVIDENCE_DST; mkdir -p $EVIDENCE_DST
cp profiles/traffic.json $EVIDENCE_DST/traffic.json
python tools/collect_evidence.py r04 final > /dev/null
python tools/collect_evidence.py r04_full final_full > /dev/null
python tools/collect_evidence.py r04_d100 final_d100 > /dev/null
python tools/collect_evidence.py r04_pl final_pl > /dev/null
# the full bench line LAST, with the counter passes just taken installed (on the box): its
# roofline block then quotes the traffic entry measured on these very kernel sources
if [ "$1" != "quick" ]; then
  cp $EVIDENCE_DST/traffic.json profiles/traffic.json
  timeout 900 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
fi
# ... and the four per-configuration lines again, for the same reason (the ones taken inside prof()
# preceded their own counter passes and quote the previous traffic entry)
requote() {  # requote <tag> <bench args...>
  TAG=$1; shift
  timeout 600 python bench.py --no-cpu-baseline --no-variants --steps 20 --warmup 4 --cross-check-seconds 0.25 $* > $EVIDENCE_DST/${TAG}_bench.json 2>/dev/null
}
if [ "$1" != "quick" ]; then
  requote r04
  requote r04_full --evaluation full
  requote r04_d100 --dim 100 --steps 10 --warmup 2
  requote r04_pl --workload pliklite --steps 8 --warmup 2
fi
[ -f gpurun_out/final_bench.json ] && cp gpurun_out/final_bench.json $EVIDENCE_DST/r04_bench_full_line.json
cp gpurun_out/final_smoke.log $EVIDENCE_DST/r04_smoke.log 2>/dev/null
rm -rf gpurun_out/final gpurun_out/final_full gpurun_out/final_d100 gpurun_out/final_pl
du -sh gpurun_out; ls $EVIDENCE_DST
