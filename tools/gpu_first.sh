set -x
cd $GRAFT_REPO_ROOT
rocminfo | grep -m2 -E "gfx|Compute Unit" 
nproc; grep -m1 "model name" /proc/cpuinfo
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25
timeout 300 python tools/quick_engine_bench.py 30 65536 64 300 2>&1 | tail -3
timeout 300 python tools/quick_engine_bench.py 30 65536 256 300 2>&1 | tail -3
timeout 300 python tools/quick_engine_bench.py 30 262144 64 300 2>&1 | tail -3
