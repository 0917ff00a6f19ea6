cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/d40; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --dim 40 --no-cpu-baseline --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/trace.log
tail -1 $OUT/bench.json | cut -c1-300
ls $OUT/trace
