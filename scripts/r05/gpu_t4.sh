# round 5: A/B of the shared heavy stream (head wide group of all passes on one queue), and the operating points (main streams)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r05_t4; mkdir -p $out
run() { tag=$1; shift
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline "$@" > $out/b_$tag.log 2> $out/b_$tag.err
  grep '^{' $out/b_$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['selfcheck'], 'p50/p99', d['latency_under_load']['ms_p50_p99'], 'win', d['config']['window_ms_min_median_max'])" || tail -3 $out/b_$tag.err; }
run h0 --heavy-streams 0
run h1 --heavy-streams 1
run h2 --heavy-streams 2
run h0b --heavy-streams 0
run h1b --heavy-streams 1
run h1beam --heavy-streams 1 --scene beam
run h0beam --heavy-streams 0 --scene beam
run s4 --streams 4
run s8 --streams 8
run s12 --streams 12
run s8p2 --streams 8 --prefetch 2
run s4p2 --streams 4 --prefetch 2
