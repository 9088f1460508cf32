# round 5: mlp_rows with resident weights (SA1's stack): parity, per-kernel time (eager 80-scene passes), pipeline rate A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r05_t31; mkdir -p $out
timeout 1500 python3 -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_timed_path_gpu.py tests/test_golden_gpu.py -m gpu -x -q 2>&1 | tail -4
A="--steps 5 --warmup 2 --batch 80 --streams 1 --no-graph --cpu-scenes 0 --no-roofline --no-legs --preroll 0 --windows 1"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/k -o k -- python3 bench.py $A > $out/k.log 2>&1
f=$(find $out/k -name "*kernel_stats.csv" | head -1); grep "mlp_rows" $f | cut -c1-150; rm -rf $out/k
export DET6D_EXPERIMENTS_LIB=1
for i in 1 2; do for r in 1 0; do
DET6D_ROWS_RESIDENT=$r timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline > $out/b_$r$i.log 2> $out/b_$r$i.err
grep '^{' $out/b_$r$i.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('resident=$r', d['value'], d['selfcheck'], d['crosscheck']['fit_scenes_per_s'])" || tail -3 $out/b_$r$i.err
done; done
