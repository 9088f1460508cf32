# round 5: 80-scene passes (merge 10 at K = 20) on the other workloads
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r05_t16; mkdir -p $out
run() { tag=$1; shift
  timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline "$@" > $out/b_$tag.log 2> $out/b_$tag.err
  grep '^{' $out/b_$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['selfcheck'], 'p50/p99', d['latency_under_load']['ms_p50_p99'], d['config']['scenes_per_pass'], 'cold', d['cold']['scenes_per_s'])" || tail -3 $out/b_$tag.err; }
run 65536_m4 --cfg synthetic_models/det6d_65536.yaml --points 65536 --batch 8 --merge 4
run 65536_m10 --cfg synthetic_models/det6d_65536.yaml --points 65536 --batch 8 --merge 10
run 65536_m5 --cfg synthetic_models/det6d_65536.yaml --points 65536 --batch 8 --merge 5
run 3class_m8 --cfg kitti_models/det6d_3class.yaml --batch 4 --merge 8
run 3class_m20 --cfg kitti_models/det6d_3class.yaml --batch 4 --merge 20
run sloped_m10 --cfg slopedkitti_models/det6d_car.yaml --tilt --merge 10
DET6D_DENSE_ROWS=1 run dense_m4 --merge 4
DET6D_DENSE_ROWS=1 run dense_m10 --merge 10
run h2d_m10 --merge 10 --h2d
