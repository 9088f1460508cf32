cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05_t17; mkdir -p $out
( timeout 2400 python3 -m pytest tests/test_timed_path_gpu.py -m gpu -q -x -k "coalesced or bench_entry" ) > $out/pytest.log 2>&1; tail -5 $out/pytest.log | cut -c1-200
export GPU_MAX_HW_QUEUES=24
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/bench_20.log 2> $out/bench_20.err
grep '^{' $out/bench_20.log | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print(d['value'], d['selfcheck'], d['config']['scenes_per_pass'], d['latency_under_load']['ms_p50_p99'], 'cold', d['cold']['scenes_per_s'])
r=d['roofline']; print('roofline', r['achieved'], r['frac'], r['kernel_ms_per_pass'], r['saturated'], r['traffic'], r['traffic_source'])
print(json.dumps(d['operating_points']))
print({k[:30]: (v.get('scenes_per_s'), v.get('latency_under_load_ms')) for k,v in d['other_configs'].items()})
print('dense', d['dense_rows'].get('scenes_per_s'), 'merge1', d['one_pass_per_batch'].get('scenes_per_s'), 'h2d', d['h2d_inclusive'].get('scenes_per_s'), 'pipeline', d['pipeline'].get('scenes_per_s'))
"; tail -4 $out/bench_20.err
