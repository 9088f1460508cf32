# round 5: which GEMM-family launches does the pass period pay for in full?  (scripts/r05/whatif_twice.py)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r05_t28; mkdir -p $out
for mode in none rows chain small linear dominant none; do
  timeout 900 python3 scripts/r05/whatif_twice.py $mode --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline > $out/b_$mode.log 2> $out/b_$mode.err
  grep '^{' $out/b_$mode.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode', d['value'], d['selfcheck'], 'fit', d['crosscheck']['fit_scenes_per_s'], 'pass period ms', round(80e3/d['value'],3))" || tail -3 $out/b_$mode.err
done
