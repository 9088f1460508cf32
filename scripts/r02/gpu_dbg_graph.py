import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np, torch
from de6d_amd.runtime import load_config, build_model, GraphedDet6D
from tests.test_model_gpu import flat_points
from tests.util import make_batch
for cfg_name, b, tilt in [('kitti_models/det6d_car.yaml', 8, False), ('slopedkitti_models/det6d_car.yaml', 8, True), ('slopedkitti_models/det6d_car.yaml', 2, True)]:
    cfg = load_config(cfg_name)
    model = build_model(cfg, seed=77, device='cuda')
    n = 16384
    runner = GraphedDet6D(model, b, n)
    for seed in (8100, 8200):
        pts = torch.from_numpy(flat_points(make_batch(seed, b, n, tilt=tilt))).cuda()
        with torch.no_grad():
            bd = {'batch_size': b, 'points': pts}
            eager, _ = model(bd)
        torch.cuda.synchronize()
        preds = runner.launch(pts).finalize()
        torch.cuda.synchronize()
        g = runner.batch_dict
        for lvl in range(3):
            same = torch.equal(g['point_coords_list'][lvl], bd['point_coords_list'][lvl])
            print(cfg_name, b, seed, 'level', lvl, 'same' if same else 'DIFF', flush=True)
        print('  boxes same', all(torch.equal(a['pred_boxes'], e['pred_boxes']) for a, e in zip(preds, eager)))
