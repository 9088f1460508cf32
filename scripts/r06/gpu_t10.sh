# round 6: pipeline A/B in the knobs build: the 128 x 64 balance rule of det6d_linear; pipeline shape around the default
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24; export DET6D_KNOBS_LIB=1
out=gpurun_out/r06_t10; mkdir -p $out
B="--gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline"
one() { tag=$1; shift; extra=""; envs=""
  for a in "$@"; do case $a in --*) extra="$extra ${a/=/ }";; *) envs="$envs $a";; esac; done
  env $envs python3 bench.py $B $extra > $out/$tag.log 2> $out/$tag.err
  grep '^{' $out/$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['selfcheck'], d['crosscheck']['fit_scenes_per_s'], d['latency_under_load']['ms_p50_p99'])" || tail -3 $out/$tag.err; }
for i in 1 2; do
one bal0_$i DET6D_LINEAR_BALANCE64=0
one bal1_$i X=1
done
one s12 X=1 --streams=12
one s20 X=1 --streams=18
one p2 X=1 --prefetch=2
one p6 X=1 --prefetch=6
one ss4 X=1 --sampler-streams=4
one ss3 X=1 --sampler-streams=3
one m16 X=1 --merge=5
one beam0 DET6D_LINEAR_BALANCE64=0 --scene=beam
one beam1 X=1 --scene=beam
