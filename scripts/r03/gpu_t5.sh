# mlp_group epilogue: vector stores + contiguous atomics: parity, A/B, per-kernel WRITE_SIZE on uniform scenes
cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=24; export TMPDIR=/tmp
python3 -m pytest tests/test_compact_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -4
python3 -m pytest tests/test_timed_path_gpu.py -x -q -m gpu -k "bench_group or ray_cast or other_baseline or dense_rows" 2>&1 | tail -4
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], d["value"], d["config"]["window_ms_min_median_max"], "cold", d["cold"]["scenes_per_s"], d["selfcheck"])'
B="--no-legs --cpu-scenes 0 --no-roofline --steps 20 --warmup 5"
for i in 1 2 3; do
python3 bench.py $B 2>/dev/null | python3 -c "$show" "uniform"
python3 bench.py $B --scene beam 2>/dev/null | python3 -c "$show" "beam"
done
out=gpurun_out/t5_write; mkdir -p $out
A="--steps 6 --warmup 2 --batch 32 --streams 1 --no-graph --cpu-scenes 0 --no-roofline --no-legs --preroll 0 --windows 1"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out -o pmc -- python3 bench.py $A > $out/log.txt 2>&1
python3 - $out <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(float); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    if r['Counter_Name'] == 'WRITE_SIZE':
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:70]
        agg[n] += float(r['Counter_Value']); cnt[n] += 1
npass = max(1, cnt[[k for k in cnt if 'pack_points' in k][0]])
for n, v in sorted(agg.items(), key=lambda x: -x[1])[:14]:
    print("%-72s %9.1f MB per pass (%d launches per pass)" % (n, v / npass / 1024.0, cnt[n] / npass))
print("total MB per pass", sum(agg.values()) / npass / 1024.0)
PY
find $out -name "*.csv" -size +2M -delete
