# round 6: score-weighted multi-pick sampler in the model: parity suites (all configurations), pipeline A/B against the previous
# commit's library (benchmark scenes, 65536-point configuration), one frame
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r06_t27; mkdir -p $out
PREV=$GRAFT_REPO_ROOT/scripts/r06/prev/libdet6d_hip_prev.so
timeout 2400 python3 -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_golden_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 2400 python3 -m pytest tests/test_timed_path_gpu.py -m gpu -x -q -k "regime or other_baseline or cooperative or full_size_vs or coalesced" 2>&1 | tail -3
B="--gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline"
one() { tag=$1; shift; extra=""; envs=""
  for a in "$@"; do case $a in --*) extra="$extra ${a/=/ }";; *) envs="$envs $a";; esac; done
  env $envs python3 bench.py $B $extra > $out/$tag.log 2> $out/$tag.err
  grep '^{' $out/$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['selfcheck'], d['crosscheck']['fit_scenes_per_s'], d['latency_b1']['ms_per_frame'], d['latency']['ms_per_batch'])" || tail -3 $out/$tag.err; }
for i in 1 2; do
one old_$i DET6D_KNOBS_LIB=$PREV
one new_$i X=1
done
for i in 1 2; do
one old_65536_$i DET6D_KNOBS_LIB=$PREV --cfg=synthetic_models/det6d_65536.yaml --points=65536 --batch=8
one new_65536_$i X=1 --cfg=synthetic_models/det6d_65536.yaml --points=65536 --batch=8
done
