#!/bin/bash
# The ONE GPU command line of this repo (replaces the per-experiment scripts of rounds 1-4).
# Run on the GPU box through gpurun, e.g.
#   gpurun --timeout 1500 -- 'bash tools/gpu.sh tests tests/test_gpu_bench_geometry.py; bash tools/gpu.sh bench'
#
#   tests [pytest args]        pytest -m gpu (default: the whole GPU suite) -> gpurun_out/tests.log
#   smoke                      __graft_entry__.smoke()
#   bench [bench args]         python bench.py ... -> gpurun_out/bench_<tag>.json (TAG=..., default "line")
#   evidence <rNN> [quick]     tests + smoke + per-configuration bench lines with rocprofv3
#                              kernel-trace stats and separate PMC passes of the SAME command
#                              (headline, evaluation: full, d = 100, plik-lite), summarised on the
#                              box by tools/collect_evidence.py into gpurun_out/evidence/<rNN>_*
#                              (copy those into profiles/), then the full bench line re-quoted on
#                              the counter passes just taken
#   evidence1 <rNN>            the headline's part of `evidence` alone (bench line, kernel trace, PMC passes)
#   ab "<v1 v2 ..>" [bench args]     same-box A/B of library builds cobaya_amd/csrc/_exp/lib_<v>.so
#                              ("cur" = the built libmcmc_hip.so), alternated REPS (3) times
#   abpmc "<v1 ..>" <kernel-like> [bench args]   SQ + LDS counters of one kernel for those builds
#   plab "<v..>" "<v..>" "<v..>"   plik-lite kernels of those builds: GPU tests / kernel times / per-phase clocks
#   pmc <kernel-like> "<counters>" -- <command>  one PMC pass of any command, per-kernel averages
#   timeline [bench args | -- <command>]   start / duration / gap of every kernel and copy around
#                              the middle step kernel of a bench run (or of any command)
#   py <script> [args]         python <script> (tools/*.py benches) -> stdout
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
BENCH_QUICK="--no-cpu-baseline --no-variants --steps 20 --warmup 4 --cross-check-seconds 0.25"
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SMEM"
LDS="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE"

libof() { [ "$1" = cur ] && echo "$PWD/cobaya_amd/csrc/libmcmc_hip.so" || echo "$PWD/cobaya_amd/csrc/_exp/lib_$1.so"; }

counters_of() {  # counters_of <dir with *_results.db> <kernel-like> <label>
  python - "$1" "$2" "$3" <<'PY'
import glob, sqlite3, sys
rows = []
for db in glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True):
    c = sqlite3.connect(db)
    rows += c.execute("select counter_name, avg(value), count(*), avg(duration) from counters_collection "
                      "where kernel_name like ? group by counter_name", (f"%{sys.argv[2]}%",)).fetchall()
print(sys.argv[3], {r[0]: float("%.6g" % r[1]) for r in rows},
      "dispatches", rows[0][2] if rows else 0, "avg_ns", round(rows[0][3]) if rows else None)
PY
}

prof() {  # prof <outdir> <bench args...>: bench line, kernel trace, PMC passes of the same command
  OUT=$1; shift
  rm -rf $OUT; mkdir -p $OUT
  CMD="python bench.py $BENCH_QUICK $*"
  echo "$CMD" > $OUT/cmd.txt
  timeout 600 $CMD > $OUT/bench.json 2> $OUT/bench.err
  rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
  rocprofv3 --pmc $SQ -d $OUT/pmc_sq -o p -- $CMD > /dev/null 2>&1
  rocprofv3 --pmc $LDS -d $OUT/pmc_lds -o p -- $CMD > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- $CMD > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- $CMD > /dev/null 2>&1
  if echo "$*" | grep -q pliklite; then
    rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES -d $OUT/pmc_mfma -o p -- $CMD > /dev/null 2>&1
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum -d $OUT/pmc_l2 -o p -- $CMD > /dev/null 2>&1
  fi
  find $OUT -type f ! -name "*_results.db" ! -name "*.json" ! -name "*.txt" ! -name "*.err" ! -name "*.log" -delete
}

sub=$1; shift
case "$sub" in
tests)
  [ $# -eq 0 ] && set -- tests
  timeout ${TEST_TIMEOUT:-2400} python -m pytest "$@" -m gpu -q -x 2>&1 | tail -${TAIL:-15} | tee gpurun_out/tests.log ;;
smoke)
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee gpurun_out/smoke.log ;;
bench)
  timeout 900 python bench.py "$@" > gpurun_out/bench_${TAG:-line}.json 2> gpurun_out/bench_${TAG:-line}.err
  echo "bench rc=$?"; tail -c 600 gpurun_out/bench_${TAG:-line}.err
  python - gpurun_out/bench_${TAG:-line}.json <<'PY'
import json, sys
lines = [json.loads(x) for x in open(sys.argv[1]) if x.startswith("{")]
b = lines[-1]
r = b["roofline"]
print("value %.4g  ms/step %.4f  kernel %s %.4f ms  frac %s  accept %.3f  certified %s  KL %s  line %d B" % (
    b["value"], b["ms_per_step"], r["kernel"], r["kernel_ms_per_launch"], r.get("frac"),
    b.get("acceptance_rate", -1), b.get("certified"), (b.get("posterior_check") or {}).get("KL"),
    max(len(x) for x in open(sys.argv[1]))))
for v in (x["bench_variant"] for x in lines if "bench_variant" in x):
    c = v.get("certificate") or {}
    if "error" in v:
        print("  %-14s ERROR %s" % (v.get("tag"), v["error"][:150])); continue
    print("  %-14s %.4g  kernel %.4f ms  ok=%s acc=%.3f KL=%s frac=%s  %s" % (v.get("tag"), v["value"],
          v.get("kernel_ms_per_launch") or -1, c.get("ok"), c.get("acceptance_rate") or -1, c.get("KL"),
          (v.get("roofline") or {}).get("frac"), (v.get("kernel") or "")[:60]))
PY
  ;;
evidence)
  RND=${1:-r05}; MODE=$2
  if [ "$MODE" != quick ]; then
    timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/final_gpu_tests.log
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1
  fi
  prof gpurun_out/final
  prof gpurun_out/final_full --evaluation full
  prof gpurun_out/final_d100 --dim 100 --steps 10 --warmup 2
  prof gpurun_out/final_pl --workload pliklite --steps 8 --warmup 2
  [ -f gpurun_out/final_gpu_tests.log ] && cp gpurun_out/final_gpu_tests.log gpurun_out/final/gpu_tests.log
  export EVIDENCE_DST=$PWD/gpurun_out/evidence EVIDENCE_COMMIT=$(cat tools/.evidence_commit 2>/dev/null)
  rm -rf $EVIDENCE_DST; mkdir -p $EVIDENCE_DST
  cp profiles/traffic.json $EVIDENCE_DST/traffic.json
  python tools/collect_evidence.py $RND final > /dev/null
  python tools/collect_evidence.py ${RND}_full final_full > /dev/null
  python tools/collect_evidence.py ${RND}_d100 final_d100 > /dev/null
  python tools/collect_evidence.py ${RND}_pl final_pl > /dev/null
  cp $EVIDENCE_DST/traffic.json profiles/traffic.json   # (on the box: the lines below quote it)
  if [ "$MODE" != quick ]; then
    timeout 900 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
    cp gpurun_out/final_bench.json $EVIDENCE_DST/${RND}_bench_full_line.json
  fi
  requote() { TAG=$1; shift; timeout 600 python bench.py $BENCH_QUICK $* > $EVIDENCE_DST/${TAG}_bench.json 2>/dev/null; }
  requote $RND
  requote ${RND}_full --evaluation full
  requote ${RND}_d100 --dim 100 --steps 10 --warmup 2
  requote ${RND}_pl --workload pliklite --steps 8 --warmup 2
  cp gpurun_out/final_smoke.log $EVIDENCE_DST/${RND}_smoke.log 2>/dev/null
  rm -rf gpurun_out/final gpurun_out/final_full gpurun_out/final_d100 gpurun_out/final_pl
  du -sh gpurun_out; ls $EVIDENCE_DST ;;
evidence1)
  # the headline's kernel-trace stats and PMC passes alone (a late change of the kernel sources:
  # traffic.json must follow, or the bench line says so)
  RND=${1:-r05}
  prof gpurun_out/final
  export EVIDENCE_DST=$PWD/gpurun_out/evidence EVIDENCE_COMMIT=$(cat tools/.evidence_commit 2>/dev/null)
  rm -rf $EVIDENCE_DST; mkdir -p $EVIDENCE_DST
  cp profiles/traffic.json $EVIDENCE_DST/traffic.json
  python tools/collect_evidence.py $RND final > /dev/null
  rm -rf gpurun_out/final
  ls $EVIDENCE_DST ;;
ab)
  VARIANTS=$1; shift
  O=gpurun_out/ab; mkdir -p $O
  for rep in $(seq 1 ${REPS:-3}); do
    for v in $VARIANTS; do
      MCMC_HIP_LIB=$(libof $v) timeout 300 python bench.py --no-cpu-baseline --no-variants --cross-check-seconds 0 "$@" > $O/b_${v}_$rep.json 2>> $O/err.log
      python - $O/b_${v}_$rep.json $v $rep <<'PY'
import json, sys
try:
    b = json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
    print(sys.argv[2], sys.argv[3], "value %.4g ms/step %.4f kernel %.4f ms (%s) accept %.3f certified %s" % (
        b["value"], b["ms_per_step"], b["roofline"]["kernel_ms_per_launch"], b["roofline"]["kernel"],
        b.get("acceptance_rate", -1), b.get("certified")))
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
    done
  done ;;
plab)
  # plab "<variants to test>" "<variants to time>" "<variants to clock>": the plik-lite kernels of
  # cobaya_amd/csrc/_exp/lib_<v>.so (tools/exp_pl_variants.sh; "cur" = the built library): the GPU
  # tests of the likelihood, tools/pliklite_bench.py twice alternated, tools/pl_clocks.py
  export MCMC_HIP_LIB_COMPAT=1
  for v in $1; do
    echo "== tests $v"; MCMC_HIP_LIB=$(libof $v) timeout 600 python -m pytest tests/test_gpu_pliklite.py -m gpu -q -x 2>&1 | tail -2
  done
  for rep in 1 2; do for v in $2; do
    echo "== $v $(MCMC_HIP_LIB=$(libof $v) timeout 300 python tools/pliklite_bench.py 26 65536 24 2>&1 | grep 'per step')"
  done; done
  for v in $3; do
    echo "== clocks $v"; MCMC_HIP_LIB=$(libof $v) timeout 200 python tools/pl_clocks.py 2>&1 | tail -42
  done ;;
abpmc)
  VARIANTS=$1; KLIKE=$2; shift 2
  O=gpurun_out/abpmc; rm -rf $O; mkdir -p $O
  for v in $VARIANTS; do
    for grp in SQ LDS; do
      MCMC_HIP_LIB=$(libof $v) timeout 300 rocprofv3 --pmc ${!grp} -d $O/p_${v}_$grp -o p -- python bench.py $BENCH_QUICK "$@" > /dev/null 2>> $O/err.log
    done
    counters_of $O "$KLIKE" $v | tee -a gpurun_out/abpmc.txt
    rm -rf $O/p_${v}_*
  done ;;
pmc)
  KLIKE=$1; CTRS=$2; shift 3
  O=gpurun_out/pmc_one; rm -rf $O; mkdir -p $O
  timeout 600 rocprofv3 --pmc $CTRS -d $O -o p -- "$@" > $O/cmd.log 2>&1
  counters_of $O "$KLIKE" "${LABEL:-pmc}" | tee -a gpurun_out/pmc.txt
  find $O -name "*.db" -delete ;;
timeline)
  O=gpurun_out/timeline; rm -rf $O; mkdir -p $O
  if [ "$1" = "--" ]; then   # timeline -- <any command>
    shift
    timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d $O/prof -o t -- "$@" > $O/cmd.log 2>&1
  else
    timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d $O/prof -o t -- python bench.py --no-variants --no-cpu-baseline --cross-check-seconds 0 "$@" > $O/bench.json 2> $O/bench.err
  fi
  python - "$O" <<'PY'
import glob, sqlite3, sys
O = sys.argv[1]
db = sqlite3.connect(glob.glob(f"{O}/prof/**/*.db", recursive=True)[0])
rows = db.execute("select start, end, name, queue_id from kernels order by start").fetchall()
try:   # (with the bytes moved where the view has them)
    mc = db.execute("select start, end, name || ' ' || size || ' B', 0 from memory_copies order by start").fetchall()
except Exception:
    try:
        mc = db.execute("select start, end, name, 0 from memory_copies order by start").fetchall()
    except Exception as e:
        print("no memory_copies view:", e); mc = []
ev = sorted(rows + mc)
idx = [i for i, r in enumerate(ev) if "step_" in r[2] or "pl_fused" in r[2]]
i0, i1 = idx[len(idx) // 2], idx[min(len(idx) // 2 + int(__import__("os").environ.get("TL_STEPS", "5")), len(idx) - 1)]
t0 = ev[i0][0]
with open(f"{O}/timeline.txt", "w") as f:
    for r in ev[i0:i1 + 1]:
        line = f"{(r[0]-t0)/1e3:10.2f} {(r[1]-t0)/1e3:10.2f} dur {(r[1]-r[0])/1e3:8.2f} us q{r[3]} {r[2][:90]}"
        f.write(line + "\n"); print(line)
PY
  rm -rf $O/prof ;;
py)
  timeout ${PY_TIMEOUT:-900} python "$@" ;;
*)
  sed -n 2,28p "$0"; exit 2 ;;
esac
