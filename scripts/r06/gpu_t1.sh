# round 6: the new timed-regime parity test + the suites the advisor fixes touch; the worker line with the clock / power sampler
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r06_t1; mkdir -p $out
( time timeout 1700 python3 -m pytest tests/test_timed_path_gpu.py -m gpu -x -q -k "regime or cooperative or stream_equals" ) 2>&1 | tail -6
( time timeout 1200 python3 -m pytest tests/test_ops_gpu.py tests/test_ball_query_shapes_gpu.py tests/test_compact_gpu.py -m gpu -x -q ) 2>&1 | tail -4
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 ) > $out/w.log 2> $out/w.err
grep '^{' $out/w.log | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('value', d['value'], d['selfcheck'], d.get('clocks'))
print({k: r[k] for k in r if not isinstance(r[k], (dict, list))})
for l in r['launches']: print(l)
print(r.get('clock_power_detail'))"; tail -3 $out/w.err
