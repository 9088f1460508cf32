# round 5: pass size again, now with the long span (16 capacities) and the two cross-checks
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r05_t22; mkdir -p $out
run() { tag=$1; shift
  timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline "$@" > $out/b_$tag.log 2> $out/b_$tag.err
  grep '^{' $out/b_$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['selfcheck'], d['crosscheck']['fit_scenes_per_s'], d['crosscheck']['whole_stream_scenes_per_s'], 'p50', d['latency_under_load']['ms_p50_p99'][0], 'windows', d['config']['windows'], 'stream s', d['stream_total_s'])" || tail -3 $out/b_$tag.err; }
for rep in 1 2; do
run m4_$rep --merge 4
run m8_$rep --merge 8
run m10_$rep --merge 10
run m16_$rep --merge 16
run m20_$rep --merge 20
run m10p6_$rep --merge 10 --prefetch 6
done
run m4beam --merge 4 --scene beam
run m10beam --merge 10 --scene beam
