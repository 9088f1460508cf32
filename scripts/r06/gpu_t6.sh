# round 6: the same A/B in the KNOBS build (shipped kernels, switches live, no instrumentation), interleaved, two repeats
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24; export DET6D_KNOBS_LIB=1
out=gpurun_out/r06_t6; mkdir -p $out
B="--gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline"
one() { tag=$1; shift
  env "$@" python3 bench.py $B > $out/$tag.log 2> $out/$tag.err
  grep '^{' $out/$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['selfcheck'], d['crosscheck']['fit_scenes_per_s'], d.get('clocks'))" || tail -3 $out/$tag.err; }
for i in 1 2; do
one r05_$i DET6D_GROUP_STREAM=2 DET6D_GROUP_SA3_WAVES=4 DET6D_LINEAR_PIPE=0
one sa3w8_$i DET6D_GROUP_STREAM=2 DET6D_GROUP_SA3_WAVES=8 DET6D_LINEAR_PIPE=0
one stream3_$i DET6D_GROUP_STREAM=3 DET6D_GROUP_SA3_WAVES=4 DET6D_LINEAR_PIPE=0
one pipe_$i DET6D_GROUP_STREAM=2 DET6D_GROUP_SA3_WAVES=4 DET6D_LINEAR_PIPE=1
one new_$i X=1
one s7_$i DET6D_GROUP_STREAM=7
one s6_$i DET6D_GROUP_STREAM=6
done
