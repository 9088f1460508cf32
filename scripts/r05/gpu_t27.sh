# round 5: paced arrivals (headway) as an operating point: latency at ~94-97 % of the saturated rate
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r05_t27; mkdir -p $out
run() { tag=$1; shift
  timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline "$@" > $out/b_$tag.log 2> $out/b_$tag.err
  grep '^{' $out/b_$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['selfcheck'], d['crosscheck']['fit_scenes_per_s'], 'p50/p99', d['latency_under_load']['ms_p50_p99'], 'window ms', d['config']['window_ms_min_median_max'], 'host blocked', d['host_blocked_frac'])" || tail -3 $out/b_$tag.err; }
run m4_h2.2_p4 --merge 4 --headway-ms 2.2
run m4_h2.25_p4 --merge 4 --headway-ms 2.25
run m4_h2.3_p3 --merge 4 --headway-ms 2.3 --prefetch 3
run m4_h2.3_p2 --merge 4 --headway-ms 2.3 --prefetch 2
run m4_h2.25_p2 --merge 4 --headway-ms 2.25 --prefetch 2
run m4_h2.35_p2 --merge 4 --headway-ms 2.35 --prefetch 2
run m4_h2.3_p2_s8 --merge 4 --headway-ms 2.3 --prefetch 2 --streams 8
run m2_h1.2_p4 --merge 2 --headway-ms 1.2
run m2_h1.2_p3 --merge 2 --headway-ms 1.2 --prefetch 3
run m2_h1.25_p4 --merge 2 --headway-ms 1.25
run m5_h2.9_p2 --merge 5 --headway-ms 2.9 --prefetch 2
run m5_h2.8_p3 --merge 5 --headway-ms 2.8 --prefetch 3
run m10_h5.5_p2 --merge 10 --headway-ms 5.5 --prefetch 2
