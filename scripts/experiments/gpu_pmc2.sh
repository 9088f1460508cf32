cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
STEPS=4; WARM=2; TOT=$((STEPS+WARM+1))
mkdir -p gpurun_out/pmc2
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc2/sq -o pmc -- python3 bench.py --steps $STEPS --warmup $WARM --streams 1 --no-graph --cpu-scenes 0 --no-roofline > gpurun_out/pmc2/sq.log 2>&1
tail -1 gpurun_out/pmc2/sq.log | cut -c1-100
python3 - <<'PY'
import csv, glob, collections
out = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('gpurun_out/pmc2/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        key = None
        for k in ('linear_kernel<128, 128', 'linear_kernel<128, 64', 'linear_kernel<64, 64', 'linear_kernel<128, 32', 'mlp_chain', 'fps_fat_kernel<9, 32', 'bq_grid_query'):
            if k in n: key = k
        if key: out[key][r['Counter_Name']] += float(r['Counter_Value'])
for k, c in out.items():
    wc = c.get('SQ_WAVE_CYCLES', 1)
    print(k, {n: round(v / wc, 4) for n, v in c.items() if n != 'SQ_WAVE_CYCLES'}, 'wave_cycles %.3g' % wc)
PY
find gpurun_out/pmc2 -name "*.csv" -size +1M -delete
