# the round's last call: the whole GPU suite, smoke() and the driver's bench command on the HEAD build
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r05_head
( time timeout 2400 python3 -m pytest tests -m gpu -q ) > gpurun_out/r05_head/pytest.log 2>&1; tail -4 gpurun_out/r05_head/pytest.log | cut -c1-200
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r05_head/bench_20.log 2> gpurun_out/r05_head/bench_20.err
grep '^{' gpurun_out/r05_head/bench_20.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['selfcheck'], d['crosscheck'], d['roofline']['frac'], d['roofline']['saturated']['frac'], d['cpu_baseline']['value'], {k:(v.get('scenes_per_s'), v.get('latency_under_load_ms')) for k,v in d['operating_points'].items()})"; tail -4 gpurun_out/r05_head/bench_20.err
