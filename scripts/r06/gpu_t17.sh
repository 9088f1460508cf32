# round 6: staggered start in the pipeline (knobs build)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24; export DET6D_KNOBS_LIB=1
out=gpurun_out/r06_t17; mkdir -p $out
B="--gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline"
one() { tag=$1; shift; extra=""; envs=""
  for a in "$@"; do case $a in --*) extra="$extra ${a/=/ }";; *) envs="$envs $a";; esac; done
  env $envs python3 bench.py $B $extra > $out/$tag.log 2> $out/$tag.err
  grep '^{' $out/$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['selfcheck'], d['crosscheck']['fit_scenes_per_s'])" || tail -3 $out/$tag.err; }
for i in 1 2; do
one st0_$i DET6D_GROUP_STAGGER=0
one st12_$i DET6D_GROUP_STAGGER=12
one st20_$i DET6D_GROUP_STAGGER=20
done
