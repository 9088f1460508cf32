# HEAD: the whole GPU suite (311 tests) and `python bench.py` with NO flags (defaults: 192 steps, 48 warmup)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r05_head2
( time timeout 2400 python3 -m pytest tests -m gpu -x -q ) > gpurun_out/r05_head2/pytest.log 2>&1; tail -4 gpurun_out/r05_head2/pytest.log | cut -c1-200
( time python3 bench.py ) > gpurun_out/r05_head2/bench_default.log 2> gpurun_out/r05_head2/bench_default.err
grep '^{' gpurun_out/r05_head2/bench_default.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['steps'], d['warmup'], d['selfcheck'], d['crosscheck']['fit_scenes_per_s'], d['config']['scenes_per_pass'], d['roofline']['frac'], d['cpu_baseline']['value'])"; tail -4 gpurun_out/r05_head2/bench_default.err
