# round 5: does a paced pipeline fill keep the passes in flight out of lock-step, and does that change the rate?
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r05_t24; mkdir -p $out
run() { tag=$1; shift
  timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline --dump-deliveries $out/d_$tag.txt "$@" > $out/b_$tag.log 2> $out/b_$tag.err
  grep '^{' $out/b_$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['selfcheck'], d['crosscheck']['fit_scenes_per_s'], d['crosscheck']['whole_stream_scenes_per_s'], 'p50/p99', d['latency_under_load']['ms_p50_p99'], 'window ms', d['config']['window_ms_min_median_max'])" || tail -3 $out/b_$tag.err; }
run m4_p0 --merge 4
run m4_p2.2 --merge 4 --headway-ms 2.2
run m4_p4.4 --merge 4 --headway-ms 4.4
run m4_p1.1 --merge 4 --headway-ms 1.1
run m10_p0 --merge 10
run m10_p5.4 --merge 10 --headway-ms 5.4
run m10_p2.7 --merge 10 --headway-ms 2.7
run m10_p10 --merge 10 --headway-ms 10
run m4_p2.2b --merge 4 --headway-ms 2.2
run m4_p0b --merge 4
