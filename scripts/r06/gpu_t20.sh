# round 6: sequencers with deferred publish: sampler suites (exactness, us per pick), pipeline A/B against the previous commit's
# library on the benchmark scenes, ray-cast scenes and the 65536-point configuration
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r06_t20; mkdir -p $out
PREV=$GRAFT_REPO_ROOT/scripts/r06/prev/libdet6d_hip_prev.so
python3 tests/gpu_scripts/fps_seq.py > $out/fps_seq.txt 2>&1; grep "b=8 n=16384 m=4096\|b=8 n=4096 m=512\|ALL" $out/fps_seq.txt
python3 tests/gpu_scripts/fps_coop.py > $out/fps_coop.txt 2>&1; tail -3 $out/fps_coop.txt
timeout 1500 python3 -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "fps or sampler or coop" 2>&1 | tail -3
B="--gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline"
one() { tag=$1; shift; extra=""; envs=""
  for a in "$@"; do case $a in --*) extra="$extra ${a/=/ }";; *) envs="$envs $a";; esac; done
  env $envs python3 bench.py $B $extra > $out/$tag.log 2> $out/$tag.err
  grep '^{' $out/$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['selfcheck'], d['crosscheck']['fit_scenes_per_s'], d['latency_b1']['ms_per_frame'] if 'latency_b1' in d else '')" || tail -3 $out/$tag.err; }
for i in 1 2 3; do
one old_$i DET6D_KNOBS_LIB=$PREV
one new_$i X=1
done
one old_beam DET6D_KNOBS_LIB=$PREV --scene=beam
one new_beam X=1 --scene=beam
one old_65536 DET6D_KNOBS_LIB=$PREV --cfg=synthetic_models/det6d_65536.yaml --points=65536 --batch=8
one new_65536 X=1 --cfg=synthetic_models/det6d_65536.yaml --points=65536 --batch=8
one old_65536b DET6D_KNOBS_LIB=$PREV --cfg=synthetic_models/det6d_65536.yaml --points=65536 --batch=8
one new_65536b X=1 --cfg=synthetic_models/det6d_65536.yaml --points=65536 --batch=8
