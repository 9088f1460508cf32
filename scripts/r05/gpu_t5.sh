# round 5: ball-query walk depth A/B (experiments build: DET6D_BQ_WALK = 1 / 2 / 4), idle-chip launch times and the pipeline
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r05_t5; mkdir -p $out
for w in 1 2 4; do
  export DET6D_EXPERIMENTS_LIB=1 DET6D_BQ_WALK=$w
  STEPS=4 NOPMC=1 bash scripts/r04/gpu_pmc.sh r05t5_w$w > $out/pmc_w$w.log 2>&1; echo "walk $w:"; grep "bq_grid_query" gpurun_out/pmc_r05t5_w$w/launches_of_one_pass.txt | cut -c1-60
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline > $out/b_w$w.log 2> $out/b_w$w.err
  grep '^{' $out/b_w$w.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  experiments lib, walk $w', d['value'], d['selfcheck'])"
done
unset DET6D_EXPERIMENTS_LIB DET6D_BQ_WALK
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline > $out/b_ship.log 2> $out/b_ship.err
grep '^{' $out/b_ship.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('shipped lib (walk 2)', d['value'], d['selfcheck'])"
for sp in "3 2" "4 1" "4 3" "5 2" "6 2" "6 3" "5 3"; do set -- $sp
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline --streams $1 --prefetch $2 > $out/b_s$1p$2.log 2> $out/b_s$1p$2.err
  grep '^{' $out/b_s$1p$2.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams $1 prefetch $2', d['value'], d['selfcheck'], 'p50/p99', d['latency_under_load']['ms_p50_p99'])"
done
