# round 6: the grid-hashed ball query for the small clouds of SA3 (1024 points) and the head (512) instead of the brute-force
# sweep (23 M vector instructions per 80-scene pass): knobs build, pipeline A/B; parity of the route against the oracle
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24; export DET6D_KNOBS_LIB=1
out=gpurun_out/r06_t13; mkdir -p $out
DET6D_GRID_MIN_N=512 timeout 900 python3 -m pytest tests/test_timed_path_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "full_size_vs or ray_cast or model" 2>&1 | tail -3
B="--gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline"
one() { tag=$1; shift; extra=""; envs=""
  for a in "$@"; do case $a in --*) extra="$extra ${a/=/ }";; *) envs="$envs $a";; esac; done
  env $envs python3 bench.py $B $extra > $out/$tag.log 2> $out/$tag.err
  grep '^{' $out/$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['selfcheck'], d['crosscheck']['fit_scenes_per_s'])" || tail -3 $out/$tag.err; }
for i in 1 2; do
one base_$i X=1
one grid1024_$i DET6D_GRID_MIN_N=1024
one grid512_$i DET6D_GRID_MIN_N=512
done
one beam_base X=1 --scene=beam
one beam_grid512 DET6D_GRID_MIN_N=512 --scene=beam
