cd $GRAFT_REPO_ROOT
for SPL in 300 600 1200; do
timeout 600 python bench.py --no-cpu-baseline --steps-per-launch $SPL --steps $((60000/SPL)) --warmup $((6000/SPL)) 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('spl $SPL bench value %.4g ms/step %.3f roofline frac %.3f kernel ms %.3f basis %.3f moments %.3f ckpts %d' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_per_launch'], d['roofline']['basis_kernel_ms_per_launch'], d['roofline']['moments_ms_per_launch'], d['config']['learn_checkpoints_in_timed_region']))"
done
