cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench value %.4g ms/step %.3f roofline frac %.3f kernel ms %.3f basis %.3f ckpts %d' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_per_launch'], d['roofline']['basis_kernel_ms_per_launch'], d['config']['learn_checkpoints_in_timed_region']))"
