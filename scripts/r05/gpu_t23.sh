# round 5: throughput over TIME within one stream (is the rate steady, or does it sag after the first second?)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r05_t23; mkdir -p $out
(for i in $(seq 1 400); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|mclk|fclk|Power' | tr '\n' ' '; echo; sleep 0.2; done) > $out/smi.log 2>&1 &
smi=$!
run() { tag=$1; shift
  timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline --dump-deliveries $out/d_$tag.txt "$@" > $out/b_$tag.log 2> $out/b_$tag.err
  grep '^{' $out/b_$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['selfcheck'], d['crosscheck']['fit_scenes_per_s'], d['crosscheck']['whole_stream_scenes_per_s'], 'p50', d['latency_under_load']['ms_p50_p99'][0], 'windows', d['config']['windows'], 'stream s', d['stream_total_s'])" || tail -3 $out/b_$tag.err
  python3 scripts/r05/delivery_rate.py $out/d_$tag.txt 8 0.25; }
run m4 --merge 4 --windows 3000
run m10 --merge 10 --windows 1200
run m4short --merge 4 --windows 192
kill $smi
sort $out/smi.log | uniq -c | sort -rn | head -12
