# round 6: list-builder granule (DET6D_COMPACT_SPLIT) and minimum class (DET6D_COMPACT_SMIN) at the 80-scene pass size, knobs build
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24; export DET6D_KNOBS_LIB=1
out=gpurun_out/r06_t12; mkdir -p $out
B="--gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline"
one() { tag=$1; shift; extra=""; envs=""
  for a in "$@"; do case $a in --*) extra="$extra ${a/=/ }";; *) envs="$envs $a";; esac; done
  env $envs python3 bench.py $B $extra > $out/$tag.log 2> $out/$tag.err
  grep '^{' $out/$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['selfcheck'], d['crosscheck']['fit_scenes_per_s'])" || tail -3 $out/$tag.err; }
one base_1 X=1
one split0 DET6D_COMPACT_SPLIT=0
one split8 DET6D_COMPACT_SPLIT=8
one split2 DET6D_COMPACT_SPLIT=2
one smin2 DET6D_COMPACT_SMIN=2
one smin4 DET6D_COMPACT_SMIN=4
one tol4 DET6D_COMPACT_TOL=4
one base_2 X=1
one beam_base X=1 --scene=beam
one beam_split0 DET6D_COMPACT_SPLIT=0 --scene=beam
one beam_split8 DET6D_COMPACT_SPLIT=8 --scene=beam
