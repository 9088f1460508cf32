# round 6: wave-private SA1 stack (mlp_rows_wave_kernel): parity, the launch replayed alone, pipeline A/B (knobs build)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r06_t11; mkdir -p $out
timeout 1500 python3 -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_golden_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 1500 python3 -m pytest tests/test_timed_path_gpu.py -m gpu -x -q -k "coalesced or full_size_vs" 2>&1 | tail -3
export DET6D_KNOBS_LIB=1
for v in 1 2; do echo "== resident=$v"; DET6D_ROWS_RESIDENT=$v python3 scripts/r06/kernel_power.py uniform 0.4 2 2>&1 | grep -v "amdgpu.ids\|^#"; done
B="--gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline"
one() { tag=$1; shift; extra=""; envs=""
  for a in "$@"; do case $a in --*) extra="$extra ${a/=/ }";; *) envs="$envs $a";; esac; done
  env $envs python3 bench.py $B $extra > $out/$tag.log 2> $out/$tag.err
  grep '^{' $out/$tag.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['selfcheck'], d['crosscheck']['fit_scenes_per_s'])" || tail -3 $out/$tag.err; }
for i in 1 2 3; do
one res1_$i DET6D_ROWS_RESIDENT=1
one res2_$i DET6D_ROWS_RESIDENT=2
done
