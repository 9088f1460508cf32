# round 5: scenes per pass revisited (the ticket scheduling and the cheaper query may have moved the optimum): --merge 4 / 6 / 8 at 24 steps
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r05_t14; mkdir -p $out
run() { tag=$1; shift
  timeout 600 python3 bench.py --gpus 1 --steps 24 --warmup 6 --worker --no-legs --cpu-scenes 0 --no-roofline "$@" > $out/b_$tag.log 2> $out/b_$tag.err
  grep '^{' $out/b_$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['selfcheck'], 'p50/p99', d['latency_under_load']['ms_p50_p99'], d['config']['scenes_per_pass'])" || tail -3 $out/b_$tag.err; }
run m4 --merge 4
run m6 --merge 6
run m8 --merge 8
run m6s12 --merge 6 --streams 12
run m8s10 --merge 8 --streams 10
run m4beam --merge 4 --scene beam
run m6beam --merge 6 --scene beam
