# round 5: s_setprio 3 for every wave of the GEMM family (experiments build, DET6D_GEMM_PRIO=1): interleaved A/B of the pipeline rate
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24 DET6D_EXPERIMENTS_LIB=1
out=gpurun_out/r05_t34; mkdir -p $out
for i in 1 2; do for pr in 0 1; do
  DET6D_GEMM_PRIO=$pr timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline > $out/b_$pr$i.log 2> $out/b_$pr$i.err
  grep '^{' $out/b_$pr$i.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('prio=$pr', d['value'], d['selfcheck'], d['crosscheck']['fit_scenes_per_s'])" || tail -3 $out/b_$pr$i.err
done; done
for pr in 0 1; do
  DET6D_GEMM_PRIO=$pr timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline --scene beam > $out/bb_$pr.log 2> $out/bb_$pr.err
  grep '^{' $out/bb_$pr.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('beam prio=$pr', d['value'], d['selfcheck'])" || tail -3 $out/bb_$pr.err
done
