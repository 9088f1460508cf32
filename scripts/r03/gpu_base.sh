# round-3 baseline: beam (ray-cast) per-launch table + profile, then A/B of uniform / beam
cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=24
bash scripts/r03/gpu_pmc.sh r03base_beam --scene beam
python3 bench.py --steps 96 --warmup 16 --worker --no-legs --cpu-scenes 0 --scene beam 2>/dev/null > gpurun_out/r03base_beam_bench.json
python3 - <<'PY'
import json
for l in open('gpurun_out/r03base_beam_bench.json'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('beam', d['value'], 'roof', r['frac'], r['kernel_ms_per_pass'], 'alg GF', r['algorithmic_gflop_per_pass'], 'sat', r['saturated'])
        for x in r['launches']: print(x)
PY
