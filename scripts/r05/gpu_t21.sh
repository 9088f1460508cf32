# round 5: the same A/B with the longer measurement span (16 pipeline capacities)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r05_t21; mkdir -p $out
run() { tag=$1; shift
  timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline "$@" > $out/b_$tag.log 2> $out/b_$tag.err
  grep '^{' $out/b_$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['selfcheck'], 'p50', d['latency_under_load']['ms_p50_p99'][0], 'windows', d['config']['windows'], 'stream s', d['stream_total_s'])" || tail -3 $out/b_$tag.err; }
for rep in 1 2; do
run ss6p4_$rep --sampler-streams 6 --prefetch 4
run ss5p5_$rep --sampler-streams 5 --prefetch 5
run ss4p5_$rep --sampler-streams 4 --prefetch 5
run ss3p5_$rep --sampler-streams 3 --prefetch 5
run ss5p4_$rep --sampler-streams 5 --prefetch 4
run ss6p5_$rep --sampler-streams 6 --prefetch 5
done
