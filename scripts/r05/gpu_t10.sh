cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r05_t10; mkdir -p $out
for a in "" "--prefetch 6" "--prefetch 8" "--scene beam"; do
timeout 600 python3 scripts/r05/stage1_latency.py --gpus 1 --steps 20 --warmup 5 --worker --no-legs --cpu-scenes 0 --no-roofline $a > $out/s1.log 2> $out/s1.err; echo "args: $a"; grep "under load\|waiting" $out/s1.err; grep '^{' $out/s1.log | cut -c90-130
done
