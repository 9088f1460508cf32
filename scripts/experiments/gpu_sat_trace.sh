cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/sat_trace; mkdir -p gpurun_out/sat_trace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sat_trace -o t -- python3 scripts/gpu_sat_trace.py > gpurun_out/sat_trace/stdout.log 2>&1
tail -1 gpurun_out/sat_trace/stdout.log
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/sat_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
fam = [r for r in rows if 'linear_kernel' in r['Kernel_Name'] or 'mlp_chain' in r['Kernel_Name']]
# last 16*6*30 launches = the saturated phase (2 warm + 4 timed replays on 16 streams)
n_sat = 16 * 6 * 30
sat, alone = fam[-n_sat:], fam[-n_sat - 30:-n_sat]
t0, t1 = min(int(r['Start_Timestamp']) for r in sat), max(int(r['End_Timestamp']) for r in sat)
print('saturated phase wall ms', (t1 - t0) / 1e6, 'launches', len(sat))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in sat:
    key = (r['Kernel_Name'].split('(')[0].replace('(anonymous namespace)::', '').replace('void ', '')[:48], r.get('Grid_Size_X', r.get('Grid_Size')))
    agg[key][0] += 1; agg[key][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
ref = {}
for r in alone:
    key = (r['Kernel_Name'].split('(')[0].replace('(anonymous namespace)::', '').replace('void ', '')[:48], r.get('Grid_Size_X', r.get('Grid_Size')))
    ref.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(v[1] for v in agg.values())
out = open('gpurun_out/sat_trace/summary.txt', 'w')
for key, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    a = ref.get(key, [0])
    out.write("%5.1f%%  avg %8.1f us under load  (alone %7.1f us)  x%d  %s grid %s\n" % (100 * us / tot, us / c, sum(a) / max(len(a), 1), c, key[0], key[1]))
out.close()
print(open('gpurun_out/sat_trace/summary.txt').read())
PY
find gpurun_out/sat_trace -name "*.csv" -delete
