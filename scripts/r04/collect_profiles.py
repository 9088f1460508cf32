"""copies the summaries of scripts/r04/gpu_final.sh from gpurun_out/ into profiles/ under their round-4 names and prints the
numbers DESIGN.md / README.md quote"""
import json
import os
import shutil
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
G, P = os.path.join(R, 'gpurun_out'), os.path.join(R, 'profiles')
SFX = sys.argv[1] if len(sys.argv) > 1 else 'z'        # gpu_pmc_all.sh suffix: 'a' = the round's first collection, 'z' = final


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
    name = 'r04%s_%s' % ('' if SFX == 'z' else SFX, tag)
    cp('pmc_r04%s_%s/kernel_stats.csv' % (SFX, tag), name + '_kernel_stats.csv')
    cp('pmc_r04%s_%s/launches_of_one_pass.txt' % (SFX, tag), name + '_launches_of_one_pass.txt')
    cp('pmc_r04%s_%s/pmc_summary.json' % (SFX, tag), name + '_pmc_summary.json')
    f = os.path.join(P, name + '_pmc_summary.json')
    if os.path.exists(f):
        s = json.load(open(f))
        print(name, 'per family:')
        for fam, r in sorted(s.get('_derived', {}).get('per_family', {}).items()):
            print('   %-24s' % fam, ' '.join('%s=%s' % (k.replace('_frac_of_wave_cycles', '').replace('_frac', ''), v) for k, v in sorted(r.items())))
if SFX != 'z':
    sys.exit(0)
line = first_json_line(os.path.join(G, 'r04_final', 'bench_20.log'))
if line:
    json.dump(line, open(os.path.join(P, 'r04_z_bench.json'), 'w'), indent=1)
    print('copied r04_z_bench.json')
for sc in ('uniform', 'beam', '65536'):
    cp('r04_pipe_%s/pipeline_kernel_stats.csv' % sc, 'r04_%s_pipeline_kernel_stats.csv' % sc)
    cp('r04_pipe_%s/trace_summary.txt' % sc, 'r04_%s_pipeline_trace_summary.txt' % sc)
    cp('r04_pipe_%s/bench_under_profiler.json' % sc, 'r04_%s_pipeline_bench_under_profiler.json' % sc)
two = first_json_line(os.path.join(G, 'r04_final', 'bench_2ranks.log'))
if two:
    json.dump(two, open(os.path.join(P, 'r04_z_bench_2ranks_one_gpu_gloo.json'), 'w'), indent=1)
if line:
    r = line['roofline']
    oc = line['other_configs']
    print('uniform', line['value'], 'ms/step', line['ms_per_step'], 'windows', line['config']['window_ms_min_median_max'])
    print('  roofline', r['achieved'], r['frac'], 'ms/pass', r['kernel_ms_per_pass'], 'sat', r['saturated'], 'traffic', r['traffic'], r['traffic_source'], 'raycast', r.get('raycast'))
    print('  merge1', line['one_pass_per_batch']['scenes_per_s'], 'cold', line['cold']['scenes_per_s'], 'latency', line['latency']['ms_per_batch'],
          'b1', line.get('latency_b1'), 'under load', line['latency_under_load']['ms_p50_p99'])
    for k, v in oc.items():
        print('  ', k[:70], v.get('scenes_per_s'), v.get('latency_under_load_ms'), v.get('fps_us_per_round'),
              {kk: v['roofline'][kk] for kk in ('achieved', 'frac', 'kernel_ms_per_pass')} if v.get('roofline') else '')
    d = line['dense_rows']
    print('dense', d['scenes_per_s'])
    print('h2d', line['h2d_inclusive']['scenes_per_s'], 'pipeline', line['pipeline']['scenes_per_s'], 'cpu', line['cpu_baseline']['value'], line['cpu_baseline']['cores'])
    print('index', line['index_kernels'])
