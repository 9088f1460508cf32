# round 5: the cached-picks what-if again, on the corrected span and the 80-scene default (item 6 quoted short-span figures)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r05_t33; mkdir -p $out
for i in 1 2; do for mode in none sa1 chain; do
  timeout 900 python3 scripts/r05/whatif_cached_fps.py $mode --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline > $out/b_$mode$i.log 2> $out/b_$mode$i.err
  grep '^{' $out/b_$mode$i.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode', d['value'], d['selfcheck'], 'fit', d['crosscheck']['fit_scenes_per_s'], 'pass period ms', round(80e3/d['value'],3))" || tail -3 $out/b_$mode$i.err
done; done
timeout 900 python3 scripts/r05/whatif_cached_fps.py sa1 --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline --scene beam > $out/b_beam_sa1.log 2> $out/b_beam_sa1.err
grep '^{' $out/b_beam_sa1.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('beam sa1', d['value'], d['selfcheck'])"
