cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -k "incremental or fuzz or posterior_moments" 2>&1 | tail -5
timeout 600 python bench.py --no-cpu-baseline --steps 40 --warmup 5 > /tmp/b.json 2> /tmp/b.err; tail -3 /tmp/b.err
python -c "
import json,sys
b=json.loads([l for l in open('/tmp/b.json') if l.startswith('{')][-1])
print('headline %.4e'%b['value'], b['roofline']['kernel'], b['roofline']['kernel_ms_per_launch'])
for v in b['variants']:
    r=v.get('roofline',{})
    print(' -', v['variant'][:50], '%.3e'%v['value'], r.get('kernel'), r.get('kernel_ms_per_launch'), v.get('kernel_ms_per_launch'))
"
