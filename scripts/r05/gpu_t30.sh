# round 5: the residency-derived grid of the row stacks: per-kernel averages (eager, 80-scene passes), parity, pipeline rate
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r05_t30; mkdir -p $out
timeout 900 python3 -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_compact_gpu.py -k "rows or mlp or model or chain or linear" -m gpu -x -q 2>&1 | tail -3
A="--steps 5 --warmup 2 --batch 80 --streams 1 --no-graph --cpu-scenes 0 --no-roofline --no-legs --preroll 0 --windows 1"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/k -o k -- python3 bench.py $A > $out/k.log 2>&1
f=$(find $out/k -name "*kernel_stats.csv" | head -1); grep "mlp_rows_kernel" $f | cut -c1-140; rm -rf $out/k
for i in 1 2; do
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline > $out/b_$i.log 2> $out/b_$i.err
grep '^{' $out/b_$i.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['selfcheck'], d['crosscheck']['fit_scenes_per_s'])"
done
