cd $GRAFT_REPO_ROOT
timeout 200 python bench.py --walkers 524288 --steps 12 --warmup 2 --no-cpu-baseline 2> /dev/null | tail -1
