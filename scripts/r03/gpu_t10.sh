cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=24
python3 -m pytest tests/test_annos_gpu.py tests/test_kitti_eval_gpu.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
python3 bench.py --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value', d['value'], 'pipeline leg', d['pipeline'])
"
done
