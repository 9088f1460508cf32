# round 6: the weight ring of the group kernels pinned with sched_barriers (four sets of 8 k-steps for TN <= 2): parity, the
# group launches replayed alone (old library = the previous commit's build), pipeline A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r06_t18; mkdir -p $out
PREV=$GRAFT_REPO_ROOT/scripts/r06/prev/libdet6d_hip_prev.so
echo "== old"; DET6D_KNOBS_LIB=$PREV python3 scripts/r06/kernel_power.py uniform 0.3 7,8,12,13 2>&1 | grep -v "amdgpu.ids\|^#"
echo "== new"; python3 scripts/r06/kernel_power.py uniform 0.3 7,8,12,13 2>&1 | grep -v "amdgpu.ids\|^#"
echo "== new, one-pass wide"; DET6D_KNOBS_LIB=1 DET6D_GROUP_STREAM=2 python3 scripts/r06/kernel_power.py uniform 0.3 13 2>&1 | grep -v "amdgpu.ids\|^#"
echo "== new, beam"; python3 scripts/r06/kernel_power.py beam 0.3 7,8,12,13 2>&1 | grep -v "amdgpu.ids\|^#"
echo "== old, beam"; DET6D_KNOBS_LIB=$PREV python3 scripts/r06/kernel_power.py beam 0.3 7,8,12,13 2>&1 | grep -v "amdgpu.ids\|^#"
timeout 1500 python3 -m pytest tests/test_compact_gpu.py tests/test_model_gpu.py tests/test_golden_gpu.py -m gpu -x -q 2>&1 | tail -3
B="--gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline"
one() { tag=$1; shift; extra=""; envs=""
  for a in "$@"; do case $a in --*) extra="$extra ${a/=/ }";; *) envs="$envs $a";; esac; done
  env $envs python3 bench.py $B $extra > $out/$tag.log 2> $out/$tag.err
  grep '^{' $out/$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['selfcheck'], d['crosscheck']['fit_scenes_per_s'])" || tail -3 $out/$tag.err; }
for i in 1 2 3; do
one old_$i DET6D_KNOBS_LIB=$PREV
one new_$i X=1
done
one old_beam DET6D_KNOBS_LIB=$PREV --scene=beam
one new_beam X=1 --scene=beam
