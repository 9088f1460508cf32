cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=24
for i in 1 2 3; do python3 bench.py --steps 20 --warmup 5 --no-legs --cpu-scenes 0 --worker 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(d['value'], 'roof', r['frac'], r['achieved'], r['kernel_ms_per_pass'], r['launches_per_pass'], r['saturated'])
"; done
