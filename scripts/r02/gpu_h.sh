cd $GRAFT_REPO_ROOT; export GPU_MAX_HW_QUEUES=24
timeout 1200 python3 -m pytest tests/test_model_gpu.py tests/test_timed_path_gpu.py -m gpu -x -q -k "graph or captured or bench_entry or group" 2>&1 | tail -4
python3 - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from de6d_amd.runtime import load_config, build_model, GraphedDet6D
from de6d_amd import synthetic
cfg = load_config('kitti_models/det6d_car.yaml'); model = build_model(cfg, seed=1234, device='cuda')
pts = torch.from_numpy(synthetic.points_tensor(synthetic.make_batch(1000, 8, 16384))).cuda()
for env in (None, '1'):
    if env: os.environ['DET6D_NO_HOIST'] = env
    r = GraphedDet6D(model, 8, 16384, points=pts)
    r.launch(); r.finalize()
    t0 = time.perf_counter()
    for _ in range(10): r.launch(); r.finalize()
    print('latency ms', 'no-hoist' if env else 'inline-hoist', (time.perf_counter() - t0) / 10 * 1e3)
PY
bash scripts/r02/gpu_beam_prof.sh
