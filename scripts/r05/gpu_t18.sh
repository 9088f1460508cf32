# round 5: pipeline shape around 80-scene passes (main streams, stages ahead, sampler streams)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r05_t18; mkdir -p $out
run() { tag=$1; shift
  timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline "$@" > $out/b_$tag.log 2> $out/b_$tag.err
  grep '^{' $out/b_$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['selfcheck'], 'p50/p99', d['latency_under_load']['ms_p50_p99'], d['config']['scenes_per_pass'])" || tail -3 $out/b_$tag.err; }
run base
run p2 --prefetch 2
run p3 --prefetch 3
run p6 --prefetch 6
run s12p3 --streams 12 --prefetch 3
run s14 --streams 14
run s18ss4 --streams 18 --sampler-streams 4
run s20ss4 --streams 20 --sampler-streams 4
run ss3 --sampler-streams 3
run ss8s14 --sampler-streams 8 --streams 14
