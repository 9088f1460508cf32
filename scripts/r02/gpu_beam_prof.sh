cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=24
python3 bench.py --steps 96 --warmup 16 --worker --no-legs --cpu-scenes 0 --scene beam 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('beam', d['value'], 'roof', r['frac'], r['kernel_ms_per_pass'], 'alg GF', r['algorithmic_gflop_per_pass'], 'sat', r['saturated'])
        for x in r['launches']: print(x)
"
