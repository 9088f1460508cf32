# round 5: (a) the query's light/heavy cut pinned (experiments build) vs adaptive, idle-chip launch times; (b) what-if: hoisted samplers cached
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r05_t6; mkdir -p $out
for c in 0 16 32 48 64 96; do
  export DET6D_EXPERIMENTS_LIB=1 DET6D_BQ_CUT=$c
  STEPS=4 NOPMC=1 bash scripts/r04/gpu_pmc.sh r05t6_c$c > $out/pmc_c$c.log 2>&1; echo "cut $c:" $(grep "bq_grid_query" gpurun_out/pmc_r05t6_c$c/launches_of_one_pass.txt | cut -c1-14 | tr '\n' ' ')
done
unset DET6D_EXPERIMENTS_LIB DET6D_BQ_CUT
for m in none sa1 chain; do
  timeout 600 python3 scripts/r05/whatif_cached_fps.py $m --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline > $out/w_$m.log 2> $out/w_$m.err
  grep '^{' $out/w_$m.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('what-if cached samplers: $m', d['value'], d['selfcheck'])" || tail -3 $out/w_$m.err
done
for m in none chain; do
  timeout 600 python3 scripts/r05/whatif_cached_fps.py $m --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline --scene beam > $out/wb_$m.log 2> $out/wb_$m.err
  grep '^{' $out/wb_$m.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('what-if cached samplers, ray-cast: $m', d['value'], d['selfcheck'])" || tail -3 $out/wb_$m.err
done
