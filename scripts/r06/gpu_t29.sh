# round 6: instruction-cache counters of the GEMM family (eager 80-scene passes): are the big group kernels (25-70 KB of code) missing
# in the 64 KB instruction cache two CUs share?
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r06_t29; mkdir -p $out
rocprofv3 -L 2>/dev/null | grep -i "icache\|ifetch\|INST_CACHE\|SQC_" | head -30 > $out/counters.txt; head -30 $out/counters.txt
A="--steps 4 --warmup 2 --batch 80 --streams 1 --no-graph --cpu-scenes 0 --no-roofline --no-legs --preroll 0 --windows 1 --worker"
for c in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  t=$(echo $c | tr ' ' '_' | cut -c1-24)
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$t -o pmc -- python3 bench.py $A > $out/$t.log 2>&1
  f=$(find $out/$t -name "*counter_collection.csv" | head -1)
  python3 - $f <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].replace('void (anonymous namespace)::', '')[:58]
    if any(s in k for s in ('mlp_group', 'linear_kernel', 'mlp_rows', 'mlp_chain', 'fps_seq')):
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
for k, v in sorted(acc.items()):
    print('%-60s' % k, ' '.join('%s=%.3g' % (c, x / max(1, n[(k, c)])) for c, x in sorted(v.items())))
PY
done
find $out -name "*.csv" -size +1M -delete
