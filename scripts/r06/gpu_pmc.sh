# per-kernel stats + PMC counters of ONE eager pass at a time (single stream): usage gpu_pmc.sh <tag> [extra bench args, e.g. --scene beam]
# rounds 4-5: two more SQ passes (instruction mix, LDS bank conflicts, wait buckets) next to FETCH / WRITE / matrix-busy
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
tag=${1:-r06}; shift; out=gpurun_out/pmc_$tag; mkdir -p $out
STEPS=${STEPS:-6}; WARM=2
A="--steps $STEPS --warmup $WARM --batch ${BATCH:-32} --streams 1 --no-graph --cpu-scenes 0 --no-roofline --no-legs --preroll 0 --windows 1 $*"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o k -- python3 bench.py $A > $out/stats.log 2>&1
grep '^{' $out/stats.log | cut -c1-300
f=$(find $out/stats -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats.csv; head -12 $f | cut -c1-160
python3 - $out/stats $out <<'PY'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
last = max(i for i, r in enumerate(rows) if 'pack_points_kernel' in r['Kernel_Name'])
with open(sys.argv[2] + '/launches_of_one_pass.txt', 'w') as f:
    for r in rows[last:]:
        n = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '')
        us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3
        f.write("%9.1f us  grid %-8s wg %-5s vgpr %-4s lds %-7s %s\n" % (us, r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')), r.get('VGPR_Count', '?'), r.get('LDS_Block_Size', '?'), n[:150]))
PY
find $out/stats -name "*kernel_trace.csv" -delete
if [ -z "$NOPMC" ]; then
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES"; do
  t=$(echo $c | tr ' ' '_' | cut -c1-24)
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$t -o pmc -- python3 bench.py $A > $out/$t.log 2>&1
  grep '^{' $out/$t.log | cut -c1-120
done
n=$(python3 - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/FETCH_SIZE/**/*counter_collection.csv', recursive=True)[0]
print(len({r['Dispatch_Id'] for r in csv.DictReader(open(f)) if 'pack_points_kernel' in r['Kernel_Name']}))
PY
)
echo passes=$n
python3 scripts/pmc_summarise.py $out $n ${BATCH:-32} > $out/pmc_summary.json; head -12 $out/pmc_summary.json
fi
find $out -name "*.csv" -size +3M -delete
