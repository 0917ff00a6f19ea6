# one iteration on the GPU box: incremental parity + fuzz, then the bench line (with and without
# the directions computed ahead)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/iter; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "incremental or moments" > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
timeout 300 python bench.py --no-cpu-baseline --no-variants > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
j=json.loads(open("gpurun_out/iter/bench.json").read().strip().splitlines()[-1]); r=j["roofline"]
print("ahead  : value %.4g ms/step %.4f kernel %.4f basis %.4f host %.4f"%(j["value"],j["ms_per_step"],r["kernel_ms_per_launch"],r["basis_kernel_ms_per_launch"],r["host_and_checkpoint_ms_per_step"]))
PY
MCMC_HIP_NO_PREFETCH=1 timeout 300 python bench.py --no-cpu-baseline --no-variants > $OUT/bench_inline.json 2>> $OUT/bench.err
python - <<'PY'
import json
j=json.loads(open("gpurun_out/iter/bench_inline.json").read().strip().splitlines()[-1]); r=j["roofline"]
print("inline : value %.4g ms/step %.4f kernel %.4f basis %.4f host %.4f"%(j["value"],j["ms_per_step"],r["kernel_ms_per_launch"],r["basis_kernel_ms_per_launch"],r["host_and_checkpoint_ms_per_step"]))
PY
timeout 300 python bench.py --no-cpu-baseline --no-variants --dim 100 --steps 10 --warmup 2 > $OUT/bench_d100.json 2>> $OUT/bench.err
python - <<'PY'
import json
j=json.loads(open("gpurun_out/iter/bench_d100.json").read().strip().splitlines()[-1]); r=j["roofline"]
print("d=100  : value %.4g ms/step %.4f kernel %.4f basis %.4f host %.4f"%(j["value"],j["ms_per_step"],r["kernel_ms_per_launch"],r["basis_kernel_ms_per_launch"],r["host_and_checkpoint_ms_per_step"]))
PY
tail -5 $OUT/bench.err
