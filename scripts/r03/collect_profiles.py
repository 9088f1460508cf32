"""copies the summaries of scripts/r03/gpu_final.sh (and of the round's baseline run) from gpurun_out/ into profiles/ under
their round-3 names, and prints the numbers DESIGN.md / README.md quote"""
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
    for line in open(path):
        if line.startswith('{'):
            return json.loads(line)


line = first_json_line(os.path.join(G, 'r03_final', 'bench_20.log'))
json.dump(line, open(os.path.join(P, 'r03_z_bench.json'), 'w'), indent=1)
print('copied r03_z_bench.json')
for tag, name in (('r03z', 'r03_z'), ('r03beam', 'r03_beam'), ('r03base_beam', 'r03_base_beam')):
    cp('pmc_%s/kernel_stats.csv' % tag, name + '_kernel_stats.csv')
    cp('pmc_%s/launches_of_one_pass.txt' % tag, name + '_launches_of_one_pass.txt')
    cp('pmc_%s/pmc_summary.json' % tag, name + '_pmc_summary.json')
cp('r03base_beam_bench.json', 'r03_base_beam_bench_worker.json')
for sc in ('uniform', 'beam'):
    cp('r03_pipe_%s/pipeline_kernel_stats.csv' % sc, 'r03_%s_pipeline_kernel_stats.csv' % sc)
    cp('r03_pipe_%s/trace_summary.txt' % sc, 'r03_%s_pipeline_trace_summary.txt' % sc)
    cp('r03_pipe_%s/bench_under_profiler.json' % sc, 'r03_%s_pipeline_bench_under_profiler.json' % sc)
two = first_json_line(os.path.join(G, 'r03_final', 'bench_2ranks.log'))
if two:
    json.dump(two, open(os.path.join(P, 'r03_z_bench_2ranks_one_gpu_gloo.json'), 'w'), indent=1)

r = line['roofline']
oc = line['other_configs']
beam = [v for k, v in oc.items() if 'ray-cast' in k and 'ramp' not in k][0]
ramp = [v for k, v in oc.items() if 'ramp' in k][0]
print('uniform', line['value'], 'ms/step', line['ms_per_step'], 'windows', line['config']['window_ms_min_median_max'], line['config']['window_ms_mean'])
print('  roofline', r['achieved'], r['frac'], 'ms/pass', r['kernel_ms_per_pass'], 'sat', r['saturated'], 'traffic', r['traffic'], r['traffic_source'])
print('  merge1', line['one_pass_per_batch']['scenes_per_s'], 'cold', line['cold']['scenes_per_s'], 'latency', line['latency']['ms_per_batch'],
      'under load', line['latency_under_load']['ms_p50_p99'])
print('beam', beam['scenes_per_s'], beam['latency_under_load_ms'], {k: beam['roofline'][k] for k in ('achieved', 'frac', 'kernel_ms_per_pass', 'algorithmic_gflop_per_pass', 'traffic')}, beam['roofline']['saturated'])
print('beam+ramp', ramp['scenes_per_s'], {k: ramp['roofline'][k] for k in ('achieved', 'frac', 'kernel_ms_per_pass', 'algorithmic_gflop_per_pass')}, ramp['roofline']['saturated'])
d = line['dense_rows']
print('dense', d['scenes_per_s'], {k: d['roofline'][k] for k in ('achieved', 'frac', 'kernel_ms_per_pass', 'algorithmic_gflop_per_pass')}, d['roofline']['saturated'])
for k, v in oc.items():
    print('  ', k[:70], v['scenes_per_s'])
print('h2d', line['h2d_inclusive']['scenes_per_s'], 'pipeline', line['pipeline']['scenes_per_s'], 'cpu', line['cpu_baseline']['value'], line['cpu_baseline']['cores'])
print('index', line['index_kernels'])
for tag in ('r03_z', 'r03_beam'):
    s = json.load(open(os.path.join(P, tag + '_pmc_summary.json')))
    g = s['_derived']['linear_kernel']
    tot_r = sum(v.get('FETCH_SIZE', 0) for k, v in s.items() if k != '_derived') * 2 * 1024
    tot_w = sum(v.get('WRITE_SIZE', 0) for k, v in s.items() if k != '_derived') * 1024
    print(tag, 'GEMM family read %.0f MB write %.0f MB per pass = %.1f MB/scene; all kernels read %.0f write %.0f = %.1f MB/scene; mlp_group write %.0f MB; MFMA busy %.3e'
          % (g['hbm_read_bytes_per_step_corrected_x2'] / 1e6, g['hbm_write_bytes_per_step'] / 1e6,
             (g['hbm_read_bytes_per_step_corrected_x2'] + g['hbm_write_bytes_per_step']) / 32e6, tot_r / 1e6, tot_w / 1e6, (tot_r + tot_w) / 32e6,
             s['mlp_group']['WRITE_SIZE'] * 1024 / 1e6, g['mfma_busy_cycles_per_step']))
