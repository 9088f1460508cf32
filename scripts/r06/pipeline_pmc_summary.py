"""Summary of ONE rocprofv3 --pmc pass over the PIPELINED bench worker (16 main + 6 sampler streams, captured graphs):
per kernel family the dispatches, counter sums, and the ratios that can be formed from them; over the whole run the
chip-level matrix-pipe busy fraction.

usage: pipeline_pmc_summary.py <dir of the rocprofv3 run> <bench json line file> > summary.json

How to read it: rocprofv3 SERIALISES dispatches while it collects counters (one kernel on the chip at a time; the kernel
trace of the same run shows it: `mean_kernels_in_flight`), so the counters describe every launch of the steady-state
stream on an otherwise idle chip, not the overlapped state.  What the pass adds over the eager one-stream PMC runs is (a) the
launches are the captured-graph replays the timed region issues, on the timed region's streams, and (b) instruction counts
(SQ_INSTS_VALU, SQ_WAVES, MFMA busy cycles) do not depend on overlap: they are the work the pipeline puts on the chip per pass.
"""
import collections
import csv
import glob
import json
import sys

root = sys.argv[1]
bench = {}
if len(sys.argv) > 2:
    for line in open(sys.argv[2]):
        if line.startswith('{'):
            bench = json.loads(line)
FAMILIES = ('linear_kernel', 'mlp_chain', 'mlp_group', 'mlp_rows', 'compact_', 'fps_coop', 'coop_', 'fps_seq_kernel', 'fps_fat_kernel', 'cell_sort',
            'ball_query_pair_kernel', 'bq_grid_query', 'bq_grid_build', 'post_', 'gather_centres', 'pack_points', 'rocprim', 'hipcub')
GEMM = ('linear_kernel', 'mlp_chain', 'mlp_group', 'mlp_rows')


def family(name):
    fam = 'other'
    for key in FAMILIES:
        if key in name:
            fam = key
    return fam


# Only the STEADY-STATE part of the run is summed: the dispatches between the 30th and the 70th percentile of the dispatch
# sequence lie inside the continuous stream of 32-scene passes (the prime, the cold run, the one-batch latency legs and the
# eager self-check passes are at the ends), so "per pass" below is per pass of the timed region
# (`scenes_per_pass` of the bench line; `whole_run_per_32_scenes` rescales to the 32-scene pass of rounds 2-4).
rows, packs = [], {}
for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r.get('Dispatch_Id') or 0), r.get('Kernel_Name', ''), r['Counter_Name'], float(r['Counter_Value'])))
        if 'pack_points_kernel' in r.get('Kernel_Name', ''):
            packs[rows[-1][0]] = int(r.get('Grid_Size') or r.get('Grid_Size_X') or 0)
ids = sorted({r[0] for r in rows})
lo_id, hi_id = ids[int(0.30 * len(ids))], ids[int(0.70 * len(ids))]
if packs:       # second half of round 5: the slice is cut on the FULL-SIZE passes (largest pack_points grid = the stream's passes),
    full = max(packs.values())      # 25th to 75th percentile: with 80-scene passes the eager 8-scene self-check passes at the end of
    fs = sorted(i for i, g in packs.items() if g == full)   # the run are a third of all dispatches and reached into the 30-70 % slice
    lo_id, hi_id = fs[len(fs) // 4], fs[3 * len(fs) // 4]
sums = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for d_id, name, ctr, val in rows:
    if not (lo_id <= d_id < hi_id):
        continue
    fam = family(name)
    sums[fam][ctr] += val
    disp[fam].add(d_id)
passes = len(disp.get('pack_points', ())) or 1
res = {'passes_profiled': passes, 'scenes_per_pass': bench.get('config', {}).get('scenes_per_pass'),
       'bench_under_profiler': {k: bench.get(k) for k in ('value', 'ms_per_step', 'selfcheck') if k in bench}}
per = {}
tot = collections.defaultdict(float)
for fam, c in sorted(sums.items()):
    d = {'dispatches_per_pass': round(len(disp[fam]) / passes, 2)}
    for k, v in c.items():
        d[k + '_per_pass'] = v / passes
        tot[k] += v
    if c.get('GRBM_GUI_ACTIVE'):
        # SQ_VALU_MFMA_BUSY_CYCLES sums 4 SIMDs x 256 CUs; GRBM_GUI_ACTIVE sums the 8 XCDs' clocks while the kernel runs
        d['matrix_pipe_busy_frac_while_running'] = round(c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / 1024.0 / (c['GRBM_GUI_ACTIVE'] / 8.0), 4)
        d['chip_cycles_per_pass'] = c['GRBM_GUI_ACTIVE'] / 8.0 / passes
    per[fam] = d
res['per_family'] = per
gemm_busy = sum(sums[f].get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) for f in GEMM)
gemm_gui = sum(sums[f].get('GRBM_GUI_ACTIVE', 0.0) for f in GEMM)
valu_gemm = sum(sums[f].get('SQ_INSTS_VALU', 0.0) for f in GEMM)
valu_rest = sum(c.get('SQ_INSTS_VALU', 0.0) for f, c in sums.items() if f not in GEMM)
res['whole_run'] = {
    'mfma_busy_cycles_per_pass_per_simd': gemm_busy / 1024.0 / passes,
    'gemm_family_chip_cycles_per_pass': gemm_gui / 8.0 / passes,
    'gemm_family_matrix_busy_frac_serialised': round(gemm_busy / 1024.0 / (gemm_gui / 8.0), 4) if gemm_gui else None,
    'all_kernels_chip_cycles_per_pass_serialised': tot.get('GRBM_GUI_ACTIVE', 0.0) / 8.0 / passes,
    'valu_wave_insts_per_pass_gemm_family_incl_mfma': valu_gemm / passes,
    'valu_wave_insts_per_pass_other_kernels': valu_rest / passes,
    'waves_per_pass': tot.get('SQ_WAVES', 0.0) / passes,
}
# the default pass is 80 scenes from the second half of round 5 on; rounds 4-5 quote these figures per 32 scenes
spp = res.get('scenes_per_pass') or 32
res['whole_run_per_32_scenes'] = {k: v * 32.0 / spp for k, v in res['whole_run'].items() if isinstance(v, float) and 'frac' not in k}
# the timed region's own clock: with the un-profiled pipeline delivering a pass every T seconds, the steady-state chip-level
# matrix-busy fraction is (busy cycles per pass per SIMD) / (T x shader clock)
res['how_to_price'] = ("steady-state matrix-pipe busy fraction = mfma_busy_cycles_per_pass_per_simd / (seconds per pass of the UN-profiled "
                       "run x shader clock); see profiles/r06_README.md for the numbers")
# overlap actually seen in this run (kernel trace of the same run, if present)
tr = glob.glob(root + '/**/*kernel_trace.csv', recursive=True)
if tr:
    rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in csv.DictReader(open(tr[0]))]
    rows.sort()
    cut = rows[len(rows) // 4: 3 * len(rows) // 4]          # the middle half: steady state
    span = (max(e for _, e in cut) - cut[0][0]) * 1e-9
    busy = sum(e - s for s, e in cut) * 1e-9
    res['trace_of_this_run'] = {'launches': len(rows), 'middle_half_window_s': span, 'sum_of_kernel_durations_s': busy,
                                'mean_kernels_in_flight': round(busy / span, 3)}
json.dump(res, sys.stdout, indent=1, sort_keys=True)
