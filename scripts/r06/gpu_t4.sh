# round 6: pipeline A/B (experiments build, interleaved, two repeats) of the knobs that won on an idle chip at the 80-scene size:
# eight waves per tile for SA3's wide group, the streaming form for the head's wide group, the software-pipelined linear slab
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r06_t4; mkdir -p $out
timeout 900 python3 -m pytest tests/test_timed_path_gpu.py -m gpu -x -q -k "regime and uniform and 16384" 2>&1 | tail -3
export DET6D_EXPERIMENTS_LIB=1
B="--gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline"
one() { tag=$1; shift
  env "$@" python3 bench.py $B > $out/$tag.log 2> $out/$tag.err
  grep '^{' $out/$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['selfcheck'], d['crosscheck']['fit_scenes_per_s'], d.get('clocks'))" || tail -3 $out/$tag.err; }
for i in 1 2; do
one base_$i X=1
one waves8_$i DET6D_GROUP_WAVES=8
one stream3_$i DET6D_GROUP_STREAM=3
one linpipe_$i DET6D_LINEAR_PIPE=1
one all3_$i DET6D_GROUP_WAVES=8 DET6D_GROUP_STREAM=3 DET6D_LINEAR_PIPE=1
done
one all3_beam DET6D_GROUP_WAVES=8 DET6D_GROUP_STREAM=3 DET6D_LINEAR_PIPE=1 DET6D_X=1
