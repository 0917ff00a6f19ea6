cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/final/gpu_tests.log
cat gpurun_out/final/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 4 2>/dev/null | tail -1 | cut -c1-200
