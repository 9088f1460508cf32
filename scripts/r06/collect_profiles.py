"""copies the summaries of scripts/r06/gpu_final.sh from gpurun_out/ into profiles/ under their round-5 names and prints the
numbers DESIGN.md / README.md / profiles/r06_README.md quote"""
import json
import os
import shutil

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
G, P = os.path.join(R, 'gpurun_out'), os.path.join(R, 'profiles')


def cp(src, dst):
    src = os.path.join(G, src)
    if os.path.exists(src):
        shutil.copyfile(src, os.path.join(P, dst))
        print('copied', dst)
    else:
        print('MISSING', src)


def first_json_line(path):
    if not os.path.exists(path):
        return None
    for line in open(path):
        if line.startswith('{'):
            return json.loads(line)


for tag in ('z', 'beam', '65536'):
    name = 'r06_%s' % tag
    cp('pmc_r06z_%s/kernel_stats.csv' % tag, name + '_kernel_stats.csv')
    cp('pmc_r06z_%s/launches_of_one_pass.txt' % tag, name + '_launches_of_one_pass.txt')
    cp('pmc_r06z_%s/pmc_summary.json' % tag, name + '_pmc_summary.json')
    f = os.path.join(P, name + '_pmc_summary.json')
    if os.path.exists(f):
        s = json.load(open(f))
        print(name, 'per family:')
        for fam, r in sorted(s.get('_derived', {}).get('per_family', {}).items()):
            print('   %-24s' % fam, ' '.join('%s=%s' % (k.replace('_frac_of_wave_cycles', '').replace('_frac', ''), v) for k, v in sorted(r.items())))
        d = s.get('_derived', {}).get('linear_kernel', {})
        print('   GEMM family HBM bytes per launch', d.get('hbm_bytes_per_launch'), 'launches', d.get('launches_per_step'), 'per scene MB',
              round(d.get('hbm_bytes_per_launch', 0) * d.get('launches_per_step', 0) / max(1, d.get('scenes_per_step', 1)) / 1e6, 2))
        tot = {k: v.get('SQ_INSTS_VALU') for k, v in s.items() if isinstance(v, dict) and 'SQ_INSTS_VALU' in v}
        print('   vector instructions per pass (incl. MFMA) by family:', {k: round(v / 1e6, 2) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])})
cp('pmc_r06z_z80/kernel_stats.csv', 'r06_z80_kernel_stats.csv')
cp('pmc_r06z_z80/launches_of_one_pass.txt', 'r06_z80_launches_of_one_pass.txt')
cp('r06_pipe_pmc/pipeline_pmc_summary.json', 'r06_pipeline_pmc_summary.json')
cp('r06_pipe_pmc/bench_under_profiler.json', 'r06_pipeline_pmc_bench_under_profiler.json')
line = first_json_line(os.path.join(G, 'r06_final', 'bench_20.log'))
if line:
    json.dump(line, open(os.path.join(P, 'r06_z_bench.json'), 'w'), indent=1)
    print('copied r06_z_bench.json')
for sc in ('uniform', 'beam', '65536', 'paced'):
    cp('r06_pipe_%s/pipeline_kernel_stats.csv' % sc, 'r06_%s_pipeline_kernel_stats.csv' % sc)
    cp('r06_pipe_%s/trace_summary.txt' % sc, 'r06_%s_pipeline_trace_summary.txt' % sc)
    cp('r06_pipe_%s/trace_overlap.txt' % sc, 'r06_%s_pipeline_trace_overlap.txt' % sc)
    cp('r06_pipe_%s/bench_under_profiler.json' % sc, 'r06_%s_pipeline_bench_under_profiler.json' % sc)
for nr in (2, 8):
    two = first_json_line(os.path.join(G, 'r06_final', 'bench_%dranks.log' % nr))
    if two:
        json.dump(two, open(os.path.join(P, 'r06_z_bench_%dranks_one_gpu_gloo.json' % nr), 'w'), indent=1)
        print(nr, 'ranks', two['value'], two['per_rank_scenes_per_s'])
for sc in ('uniform', 'beam'):
    cp('r06_final/kernel_power_%s.txt' % sc, 'r06_kernel_power_%s.txt' % sc)
f = os.path.join(P, 'r06_pipeline_pmc_summary.json')
if os.path.exists(f):
    s = json.load(open(f))
    print('pipelined PMC pass:', json.dumps(s['whole_run'], indent=1), s.get('trace_of_this_run'), s.get('passes_profiled'))
    for fam, c in sorted(s['per_family'].items(), key=lambda kv: -kv[1].get('SQ_INSTS_VALU_per_pass', 0)):
        print('   %-26s disp %5.1f  VALU %7.2f M  MFMA busy %8.1f M cyc  busy-while-running %s' % (
            fam, c['dispatches_per_pass'], c.get('SQ_INSTS_VALU_per_pass', 0) / 1e6, c.get('SQ_VALU_MFMA_BUSY_CYCLES_per_pass', 0) / 1e6,
            c.get('matrix_pipe_busy_frac_while_running')))
if line:
    r = line['roofline']
    oc = line['other_configs']
    print('uniform', line['value'], 'ms/step', line['ms_per_step'], 'windows', line['config']['window_ms_min_median_max'])
    print('  roofline', r['achieved'], r['frac'], 'ms/pass', r['kernel_ms_per_pass'], 'sat', r['saturated'], 'traffic', r['traffic'], r['traffic_source'],
          r.get('traffic_detail'), 'raycast', r.get('raycast'))
    print('  dominant', r.get('dominant_launch'))
    print('  merge1', line['one_pass_per_batch']['scenes_per_s'], 'cold', line['cold']['scenes_per_s'], 'latency', line['latency']['ms_per_batch'],
          'b1', line.get('latency_b1'), 'under load', line['latency_under_load']['ms_p50_p99'])
    print('  operating points', json.dumps(line.get('operating_points'), indent=1))
    for k, v in oc.items():
        print('  ', k[:70], v.get('scenes_per_s'), v.get('latency_under_load_ms'), v.get('fps_us_per_round'),
              {kk: v['roofline'][kk] for kk in ('achieved', 'frac', 'kernel_ms_per_pass')} if v.get('roofline') else '',
              v['roofline'].get('saturated') if v.get('roofline') else '')
    d = line['dense_rows']
    print('dense', d['scenes_per_s'], d.get('roofline', {}).get('frac'))
    print('h2d', line['h2d_inclusive']['scenes_per_s'], 'pipeline', line['pipeline']['scenes_per_s'], 'cpu', line['cpu_baseline']['value'], line['cpu_baseline']['cores'])
    print('index', line['index_kernels'])
    print('launches', r.get('launches'))
