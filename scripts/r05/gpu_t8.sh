cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r05_t8; mkdir -p $out
timeout 600 python3 scripts/r05/stage1_latency.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline > $out/s1.log 2> $out/s1.err; grep "under load" $out/s1.err; grep '^{' $out/s1.log | cut -c1-130
run() { tag=$1; shift
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline "$@" > $out/b_$tag.log 2> $out/b_$tag.err
  grep '^{' $out/b_$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['selfcheck'], 'p50/p99', d['latency_under_load']['ms_p50_p99'])" || tail -3 $out/b_$tag.err; }
run p4
run p6 --prefetch 6
run p8 --prefetch 8
run p6s8 --prefetch 6 --sampler-streams 8 --streams 14
run p3 --prefetch 3
