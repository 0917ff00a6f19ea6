# Scratch command line of the last quick GPU check (edit freely; run with
#   gpurun --timeout 600 -- 'bash tools/gpu_quick.sh > gpurun_out/quick.log 2>&1')
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 120 python tools/quick_engine_bench.py 30 65536 256 1200 2>&1 | tail -1
