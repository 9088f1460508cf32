cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/launch_list; mkdir -p gpurun_out/launch_list
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/launch_list -o t -- python3 scripts/gpu_launch_list.py > gpurun_out/launch_list/stdout.log 2>&1
python3 - <<'PY'
import csv, glob, re
f = glob.glob('gpurun_out/launch_list/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'pack_points_kernel' in n]
seq = rows[idx[-1]:]
out = open('gpurun_out/launch_list/one_pass.txt', 'w')
for r in seq:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    n = re.sub(r'at::native::', '', n)
    out.write("%8.1f us  grid %-8s wg %-5s %s\n" % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '')), n[:150]))
out.close()
print(len(seq), 'launches in one pass')
PY
find gpurun_out/launch_list -name "*.csv" -delete
