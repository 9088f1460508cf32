# round 6: HEAD check: the whole GPU suite and `python bench.py` with NO flags (192 steps, 64-scene passes) on the round's last commit
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r06_head; mkdir -p $out
( time timeout 2700 python3 -m pytest tests -m gpu -q ) > $out/pytest.log 2>&1; tail -3 $out/pytest.log | cut -c1-200
( time python3 bench.py ) > $out/bench_default.log 2> $out/bench_default.err
grep '^{' $out/bench_default.log | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('default flags: value', d['value'], d['selfcheck'], 'steps', d['steps'], 'scenes per pass', d['config']['scenes_per_pass'], 'frac', r['frac'], 'sclk', r.get('sclk_mhz'), 'b1', d['latency_b1']['ms_per_frame'], 'cfg5', r.get('cfg5_scenes_per_s'), 'raycast', r.get('raycast_scenes_per_s'))"; tail -4 $out/bench_default.err
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
