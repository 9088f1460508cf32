# where do the SA1 / SA2 chain kernels spend their wave-cycles on ray-cast scenes?  (13-30 % MFMA utilisation)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=24
out=gpurun_out/chain_pmc; mkdir -p $out
A="--steps 3 --warmup 1 --streams 1 --no-graph --cpu-scenes 0 --no-roofline --no-legs --preroll 0 --windows 1 --worker --scene beam"
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES"; do
  t=$(echo $c | tr ' ' '_' | cut -c1-30)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$t -o pmc -- python3 bench.py $A > $out/$t.log 2>&1
  tail -1 $out/$t.log | cut -c1-80
done
python3 - $out <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        fam = None
        for key in ('mlp_chain_reg_kernel<32', 'mlp_chain_reg_kernel<16', 'mlp_chain_wide_kernel<64', 'mlp_chain_wide_kernel<96', 'mlp_group_kernel<256, 512', 'mlp_group_kernel<128, 256'):
            if key in k: fam = key
        if fam is None: continue
        acc[fam][r['Counter_Name']] += float(r['Counter_Value']); n[(fam, r['Counter_Name'])].add(r['Dispatch_Id'])
for fam, c in acc.items():
    print(fam)
    for name, v in sorted(c.items()):
        print('   %-34s %14.0f per launch' % (name, v / max(len(n[(fam, name)]), 1)))
PY
find $out -name "*.csv" -size +1M -delete
