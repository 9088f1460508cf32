"""The N > 1 path on CPU: world_size-2 and world_size-8 gloo processes shard scenes, run the (oracle) pipeline on
their shard and merge detections with no data-path collective; plus the flat-bucket gradient
all-reduce used only by the optional training step."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, num_scenes, out_dir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from de6d_amd import parallel
    from de6d_amd.runtime import load_config, build_model
    from oracle import model as omodel
    from tests.util import make_batch
    cfg = load_config('synthetic_models/det6d_tiny.yaml')
    model = build_model(cfg, seed=3)
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    mine = parallel.scene_shard(num_scenes, rank, world)
    n = 1024
    results = []
    for s in mine:   # one scene per pass: scenes are independent units
        pts = np.concatenate([np.zeros((n, 1), np.float32), make_batch(500 + s, 1, n)[0]], 1)
        out = omodel.forward(cfg.MODEL, sd, pts, 1)
        results.append({'scene': s, 'boxes': out['pred_dicts'][0]['pred_boxes']})
    merged = parallel.gather_detections(results, num_scenes)
    # optional training-side collective: every rank ends with the mean gradient
    lin = torch.nn.Linear(4, 3)
    lin.weight.grad = torch.full_like(lin.weight, float(rank + 1))
    lin.bias.grad = torch.full_like(lin.bias, float(10 * (rank + 1)))
    nred = parallel.allreduce_gradients(lin.parameters())
    assert nred == 15
    mean_rank = (world + 1) / 2.0
    assert torch.allclose(lin.weight.grad, torch.full_like(lin.weight, mean_rank))
    assert torch.allclose(lin.bias.grad, torch.full_like(lin.bias, 10.0 * mean_rank))
    if rank == 0:
        np.save(os.path.join(out_dir, 'order.npy'), np.array([m['scene'] for m in merged]))
        np.save(os.path.join(out_dir, 'nbox.npy'), np.array([len(m['boxes']) for m in merged]))
    dist.barrier()
    dist.destroy_process_group()


def test_scene_shard_covers_all_scenes():
    from de6d_amd.parallel import scene_shard
    for num, world in [(8, 2), (5, 2), (7, 4), (3, 8), (32, 8)]:
        shards = [scene_shard(num, r, world) for r in range(world)]
        assert len({len(s) for s in shards}) == 1
        assert set(sum(shards, [])) == set(range(num))


@pytest.mark.parametrize("world,num_scenes", [(2, 5), (8, 13)])
def test_gloo_sharding(tmp_path, oracle_ops, world, num_scenes):
    """world 2 / 5 scenes, and the shape the driver's 8-GPU node would run: 8 ranks / 13 scenes (two scenes per rank, the
    last three slots wrap around to scenes 0..2 and are dropped by the merge)"""
    mp.spawn(_worker, args=(world, _free_port(), num_scenes, str(tmp_path)), nprocs=world, join=True)
    order = np.load(os.path.join(str(tmp_path), 'order.npy'))
    np.testing.assert_array_equal(order, np.arange(num_scenes))
    assert np.load(os.path.join(str(tmp_path), 'nbox.npy')).shape == (num_scenes,)
