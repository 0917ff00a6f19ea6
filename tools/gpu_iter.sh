# one iteration on the GPU box: incremental parity + fuzz, then the bench line (with and without
# the directions computed ahead)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/iter; rm -rf $OUT; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q ${TEST_K:+-k "$TEST_K"} > $OUT/tests.log 2>&1
tail -15 $OUT/tests.log | cut -c1-400
line() {
python - "$1" "$2" <<'PY'
import json, sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=j["roofline"]
print("%-8s: value %.4g ms/step %.4f kernel %.4f basis %.4f host %.4f  %s"%(sys.argv[2], j["value"],j["ms_per_step"],r["kernel_ms_per_launch"],r["basis_kernel_ms_per_launch"],r["host_and_checkpoint_ms_per_step"], r["kernel"]))
PY
}
timeout 300 python bench.py --no-cpu-baseline --no-variants > $OUT/bench.json 2> $OUT/bench.err; line $OUT/bench.json ahead
if [ -n "$WITH_INLINE" ]; then MCMC_HIP_NO_PREFETCH=1 timeout 300 python bench.py --no-cpu-baseline --no-variants > $OUT/bench_inline.json 2>> $OUT/bench.err; line $OUT/bench_inline.json inline; fi
timeout 300 python bench.py --no-cpu-baseline --no-variants --dim 100 --steps 10 --warmup 2 > $OUT/bench_d100.json 2>> $OUT/bench.err; line $OUT/bench_d100.json d=100
