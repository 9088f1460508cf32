# round 5: scenes per pass at the driver's K = 20 (merge must divide K): 4 (32 scenes, the default so far), 5, 10, 20; fewer streams with larger passes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r05_t15; mkdir -p $out
run() { tag=$1; shift
  timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline "$@" > $out/b_$tag.log 2> $out/b_$tag.err
  grep '^{' $out/b_$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['selfcheck'], 'p50/p99', d['latency_under_load']['ms_p50_p99'], d['config']['scenes_per_pass'], 'cold', d['cold']['scenes_per_s'])" || tail -3 $out/b_$tag.err; }
run m4 --merge 4
run m5 --merge 5
run m10 --merge 10
run m20 --merge 20
run m10s12 --merge 10 --streams 12
run m10s8 --merge 10 --streams 8
run m10s8p2 --merge 10 --streams 8 --prefetch 2
run m10beam --merge 10 --scene beam
run m10s8beam --merge 10 --streams 8 --scene beam
