cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "big_dimension" 2>&1 | tail -2
run() { timeout 120 python tools/quick_engine_bench.py "$@" 2>&1 | tail -1 | cut -c1-120; }
for d in 56 72 88 100 112 120; do run $d 65536 256 $((d*8)); done
