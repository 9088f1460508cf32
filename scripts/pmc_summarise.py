"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel family (sum over dispatches / steps)."""
import csv, glob, json, sys, collections
root, steps = sys.argv[1], int(sys.argv[2])
out = collections.defaultdict(lambda: collections.defaultdict(float))
ndisp = collections.defaultdict(set)
for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get('Kernel_Name', '')
        fam = 'other'
        for key in ('linear_kernel', 'fps_fat_kernel', 'ball_query_pair_kernel', 'post_', 'gather_rows', 'pack_points'):
            if key in name:
                fam = key
        out[fam][r['Counter_Name']] += float(r['Counter_Value'])
        ndisp[(fam, r['Counter_Name'])].add(r.get('Dispatch_Id'))
res = {}
for fam, ctrs in out.items():
    res[fam] = {c: v / steps for c, v in ctrs.items()}
    res[fam]['dispatches_per_step'] = max(len(ndisp[(fam, c)]) for c in ctrs) / steps
print(json.dumps(res, indent=1, sort_keys=True))
