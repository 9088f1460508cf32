# round 5: repeated, interleaved A/B of sampler streams / stages ahead with 80-scene passes (single runs differ by +-4 %)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r05_t20; mkdir -p $out
run() { tag=$1; shift
  timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline "$@" > $out/b_$tag.log 2> $out/b_$tag.err
  grep '^{' $out/b_$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['selfcheck'], 'p50', d['latency_under_load']['ms_p50_p99'][0], 'win', d['config']['window_ms_min_median_max'])" || tail -3 $out/b_$tag.err; }
for rep in 1 2 3; do
run ss6p4_$rep --sampler-streams 6 --prefetch 4
run ss4p4_$rep --sampler-streams 4 --prefetch 4
run ss3p5_$rep --sampler-streams 3 --prefetch 5
run ss4p5_$rep --sampler-streams 4 --prefetch 5
run ss5p5_$rep --sampler-streams 5 --prefetch 5
run ss4p6_$rep --sampler-streams 4 --prefetch 6
done
