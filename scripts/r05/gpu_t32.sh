# round 5: paced (open-loop) operating points on ray-cast scenes and on 65536-point scenes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r05_t32; mkdir -p $out
run() { tag=$1; shift
  timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline "$@" > $out/b_$tag.log 2> $out/b_$tag.err
  grep '^{' $out/b_$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['selfcheck'], d['crosscheck']['fit_scenes_per_s'], 'p50/p99', d['latency_under_load']['ms_p50_p99'], 'window ms', d['config']['window_ms_min_median_max'], 'host blocked', d['host_blocked_frac'])" || tail -3 $out/b_$tag.err; }
run beam_m10_closed --scene beam
run beam_m10_h13.0_p2 --scene beam --headway-ms 13.0 --prefetch 2
run beam_m10_h13.4_p2 --scene beam --headway-ms 13.4 --prefetch 2
run beam_m4_h5.4_p2 --scene beam --merge 4 --headway-ms 5.4 --prefetch 2
run beam_m4_h5.6_p2 --scene beam --merge 4 --headway-ms 5.6 --prefetch 2
B="--cfg synthetic_models/det6d_65536.yaml --points 65536 --batch 8"
run big_closed $B
run big_h25.4_p2 $B --headway-ms 25.4 --prefetch 2
run big_h26.5_p2 $B --headway-ms 26.5 --prefetch 2
run big_m2_h13.0_p2 $B --merge 2 --headway-ms 13.0 --prefetch 2
