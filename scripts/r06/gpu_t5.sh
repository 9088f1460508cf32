# round 6: new defaults (streaming form for the head's wide group, eight waves for SA3's wide group, pipelined linear slab) in
# the SHIPPED library: parity suites + the worker line twice; experiments build: the streaming form for SA3's wide group too
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r06_t5; mkdir -p $out
( timeout 1700 python3 -m pytest tests/test_compact_gpu.py tests/test_model_gpu.py tests/test_golden_gpu.py -m gpu -x -q 2>&1 | tail -3 )
( timeout 1700 python3 -m pytest tests/test_timed_path_gpu.py -m gpu -x -q -k "coalesced or regime or full_size_vs" 2>&1 | tail -3 )
B="--gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline"
one() { tag=$1; shift
  extra=""; for a in "$@"; do case $a in --*) extra="$extra $a";; esac; done
  env $(for a in "$@"; do case $a in --*) ;; *) echo $a;; esac; done) python3 bench.py $B $extra > $out/$tag.log 2> $out/$tag.err
  grep '^{' $out/$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['selfcheck'], d['crosscheck']['fit_scenes_per_s'], d.get('clocks'))" || tail -3 $out/$tag.err; }
one shipped_1 X=1
one shipped_2 X=1
one shipped_beam X=1 --scene=beam
export DET6D_EXPERIMENTS_LIB=1
for i in 1 2; do
one exp_s3_$i DET6D_GROUP_STREAM=3
one exp_s7_$i DET6D_GROUP_STREAM=7
done
