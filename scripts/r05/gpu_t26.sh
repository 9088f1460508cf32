# round 5: headway control (consecutive GEMM-stage launches kept apart) against the lock-step of the passes in flight
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r05_t26; mkdir -p $out
run() { tag=$1; shift
  timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline --dump-deliveries $out/d_$tag.txt "$@" > $out/b_$tag.log 2> $out/b_$tag.err
  grep '^{' $out/b_$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['selfcheck'], d['crosscheck']['fit_scenes_per_s'], d['crosscheck']['whole_stream_scenes_per_s'], 'p50/p99', d['latency_under_load']['ms_p50_p99'], 'window ms', d['config']['window_ms_min_median_max'], 'host blocked', d['host_blocked_frac'])" || tail -3 $out/b_$tag.err; }
run m4_h0 --merge 4
run m4_h1.6 --merge 4 --headway-ms 1.6
run m4_h1.9 --merge 4 --headway-ms 1.9
run m4_h2.1 --merge 4 --headway-ms 2.1
run m4_h2.3 --merge 4 --headway-ms 2.3
run m10_h0 --merge 10
run m10_h4 --merge 10 --headway-ms 4.0
run m10_h4.8 --merge 10 --headway-ms 4.8
run m10_h5.2 --merge 10 --headway-ms 5.2
run m10_h5.6 --merge 10 --headway-ms 5.6
run m4_h2.0_s8 --merge 4 --headway-ms 2.0 --streams 8
run m4_h0_s8 --merge 4 --streams 8
