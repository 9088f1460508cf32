#!/bin/bash
# pipelined trace of the 65536-point leg: which kernels are on the chip when, per stream
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/r04_pipe_65536; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o pipe -- python3 bench.py --steps 40 --warmup 8 --cpu-scenes 0 --no-roofline --no-legs --cfg synthetic_models/det6d_65536.yaml --points 65536 --batch 8 > $out/bench_stdout.log 2>&1
grep '^{' $out/bench_stdout.log > $out/bench_under_profiler.json; cut -c1-200 $out/bench_under_profiler.json
f=$(find $out -name "*kernel_stats.csv" | head -1); cp $f $out/pipeline_kernel_stats.csv
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:14]:
    print('%-60s calls %5s avg %9.1f us  %5.1f %%' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
PY
t=$(find $out -name "*kernel_trace.csv" | head -1)
python3 scripts/r02/trace_summary.py $t > $out/trace_summary.txt; head -30 $out/trace_summary.txt
python3 - $t <<'PY'
# busy structure: per 5 ms slice of the steady state, how many coop / s-fps / GEMM kernels overlap
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', r.get('Stream_Id', '?'))) for r in rows]
t0 = min(e[0] for e in ev); t1 = max(e[1] for e in ev)
mid = t0 + (t1 - t0) * 0.6
win = [e for e in ev if e[1] > mid and e[0] < mid + 60e6]
fam = lambda n: 'coop' if 'fps_coop' in n else 'sfps' if 'fps_fat' in n else 'seq' if 'fps_seq' in n else 'gemm' if ('mlp_' in n or 'linear_kernel' in n) else 'sort' if 'rocprim' in n or 'coop_keys' in n or 'group_order' in n else 'other'
for k in range(12):
    a = mid + k * 5e6; b = a + 5e6
    acc = collections.defaultdict(float)
    for s, e, n, q in win:
        o = min(e, b) - max(s, a)
        if o > 0:
            acc[fam(n)] += o / 5e6
    print('slice %2d: ' % k + '  '.join('%s %.2f' % (f, acc[f]) for f in ('coop', 'seq', 'sfps', 'gemm', 'sort', 'other')))
qs = collections.Counter(q for s, e, n, q in win if 'fps_coop' in n)
print('coop launches per queue in the window:', dict(qs))
PY
rm -f $t
