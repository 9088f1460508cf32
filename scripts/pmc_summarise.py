"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel family (sum over dispatches / steps).

usage: pmc_summarise.py <dir with one sub-directory per --pmc pass> <steps profiled> [scenes per step]
(a "step" here is one eager pass of the profiled run: --batch scenes; 32 = the launches of bench.py's coalesced passes)
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  Per MI355X_MICROARCH.md (HBM section) gfx950's
FETCH_SIZE tallies 128-byte requests at 64 bytes for 16 B/lane streaming reads, so reads are doubled;
WRITE_SIZE is exact for 16 B/lane stores.  The `_derived.linear_kernel` block is what bench.py reads
for roofline.traffic (HBM bytes per GEMM-family launch).
"""
import csv, glob, json, sys, collections
root, steps = sys.argv[1], int(sys.argv[2])
scenes = int(sys.argv[3]) if len(sys.argv) > 3 else 8
FAMILIES = ('linear_kernel', 'mlp_chain', 'mlp_group', 'mlp_rows', 'group_expand', 'compact_groups', 'fps_coop', 'fps_skip_kernel', 'fps_seq_kernel', 'fps_fat_kernel', 'ball_query_pair_kernel', 'bq_grid', 'post_',
            'gather_rows', 'pack_points')
out = collections.defaultdict(lambda: collections.defaultdict(float))
ndisp = collections.defaultdict(set)
for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get('Kernel_Name', '')
        fam = 'other'
        for key in FAMILIES:
            if key in name:
                fam = key
        out[fam][r['Counter_Name']] += float(r['Counter_Value'])
        ndisp[(fam, r['Counter_Name'])].add(r.get('Dispatch_Id'))
res = {}
for fam, ctrs in out.items():
    res[fam] = {c: v / steps for c, v in ctrs.items()}
    res[fam]['dispatches_per_step'] = max(len(ndisp[(fam, c)]) for c in ctrs) / steps
gemm = [res[f] for f in ('linear_kernel', 'mlp_chain', 'mlp_group', 'mlp_rows') if f in res]
if gemm and all('FETCH_SIZE' in g and 'WRITE_SIZE' in g for g in gemm):
    rd = sum(g['FETCH_SIZE'] for g in gemm) * 1024.0 * 2.0
    wr = sum(g['WRITE_SIZE'] for g in gemm) * 1024.0
    launches = sum(g['dispatches_per_step'] for g in gemm)
    d = {'scenes_per_step': scenes, 'launches_per_step': launches, 'hbm_read_bytes_per_step_corrected_x2': rd, 'hbm_write_bytes_per_step': wr,
         'hbm_bytes_per_launch': (rd + wr) / launches,
         'note': 'GEMM family = linear_kernel + mlp_chain_{reg,wide,}_kernel + mlp_group_kernel + mlp_rows_kernel; rocprofv3 --pmc, separate passes for FETCH_SIZE / '
                 'WRITE_SIZE / SQ counters; bench.py --streams 1 --no-graph; FETCH_SIZE doubled per '
                 'MI355X_MICROARCH.md (gfx950 counts 64 B per 128 B request for 16 B/lane reads)'}
    busy = sum(g.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) for g in gemm)
    sq = sum(g.get('SQ_BUSY_CYCLES', 0.0) for g in gemm)
    if busy and sq:
        # SQ_BUSY_CYCLES is summed over 32 SEs-worth of SQ instances x XCDs; MFMA busy over 4 SIMDs x 256 CUs
        d['mfma_busy_cycles_per_step'] = busy
        d['sq_busy_cycles_per_step'] = sq
    res['_derived'] = {'linear_kernel': d}
# per-family ratios (round 4): how busy the matrix pipe is, what the vector ALU issues beside it, what the LDS costs
ratios = {}
for fam, c in res.items():
    if fam == '_derived':
        continue
    r = {}
    if c.get('GRBM_GUI_ACTIVE') and c.get('SQ_VALU_MFMA_BUSY_CYCLES') is not None:
        # SQ_VALU_MFMA_BUSY_CYCLES sums 4 SIMDs x 256 CUs; GRBM_GUI_ACTIVE sums the 8 XCDs' clocks while the kernel runs
        r['matrix_pipe_busy_frac'] = round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / (c['GRBM_GUI_ACTIVE'] / 8.0), 4)
    if c.get('SQ_INSTS_MFMA'):
        r['valu_insts_per_mfma'] = round((c.get('SQ_INSTS_VALU', 0.0) - c['SQ_INSTS_MFMA']) / c['SQ_INSTS_MFMA'], 3)
        r['lds_insts_per_mfma'] = round(c.get('SQ_INSTS_LDS', 0.0) / c['SQ_INSTS_MFMA'], 3)
    if c.get('SQ_LDS_IDX_ACTIVE'):
        r['lds_bank_conflict_frac_of_lds_cycles'] = round(c.get('SQ_LDS_BANK_CONFLICT', 0.0) / c['SQ_LDS_IDX_ACTIVE'], 4)
    if c.get('SQ_WAVE_CYCLES'):
        for k, name in (('SQ_WAIT_INST_LDS', 'wait_inst_lds'), ('SQ_WAIT_INST_ANY', 'wait_inst_any'), ('SQ_WAIT_ANY', 'wait_any'),
                        ('SQ_ACTIVE_INST_ANY', 'active_inst_any'), ('SQ_ACTIVE_INST_VALU', 'active_inst_valu'),
                        ('SQ_ACTIVE_INST_LDS', 'active_inst_lds')):
            if k in c:
                r[name + '_frac_of_wave_cycles'] = round(c[k] / c['SQ_WAVE_CYCLES'], 4)
    if r:
        ratios[fam] = r
if ratios:
    res.setdefault('_derived', {})['per_family'] = ratios
print(json.dumps(res, indent=1, sort_keys=True))
