# round 5: sampler streams around 80-scene passes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r05_t19; mkdir -p $out
run() { tag=$1; shift
  timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline "$@" > $out/b_$tag.log 2> $out/b_$tag.err
  grep '^{' $out/b_$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['selfcheck'], 'p50/p99', d['latency_under_load']['ms_p50_p99'], d['config']['scenes_per_pass'])" || tail -3 $out/b_$tag.err; }
run ss1 --sampler-streams 1
run ss2 --sampler-streams 2
run ss3 --sampler-streams 3
run ss4 --sampler-streams 4
run ss5 --sampler-streams 5
run ss6 --sampler-streams 6
run ss2p3 --sampler-streams 2 --prefetch 3
run ss3p5 --sampler-streams 3 --prefetch 5
run ss3s18 --sampler-streams 3 --streams 18
run ss3beam --sampler-streams 3 --scene beam
run ss6beam --sampler-streams 6 --scene beam
run ss3m4 --sampler-streams 3 --merge 4
