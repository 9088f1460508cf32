# round 6 checkpoint: the whole GPU suite + the driver's bench command (every leg)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r06_full; mkdir -p $out
( time timeout 2700 python3 -m pytest tests -m gpu -q -x ) > $out/pytest.log 2>&1; tail -5 $out/pytest.log | cut -c1-300
export GPU_MAX_HW_QUEUES=24
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/bench_20.log 2> $out/bench_20.err
grep '^{' $out/bench_20.log | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('value', d['value'], d['selfcheck'], d['crosscheck'])
print({k: r[k] for k in r if not isinstance(r[k], (dict, list))})
print({k: (v.get('scenes_per_s'), v.get('selfcheck')) for k, v in d['other_configs'].items()})
print(d['operating_points']); print(d['latency_b1'], d['latency']); print(d['cpu_baseline'])"; tail -4 $out/bench_20.err
