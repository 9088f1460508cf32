"""DataProcessor for the point pipelines the Det6D configs use, running on the GPU.

Mirror of core/pcdet/datasets/processor/data_processor.py (constructor arguments, the
`DATA_PROCESSOR` YAML list, `forward(data_dict)`), restricted to the three processors the point
models configure:

    mask_points_and_boxes_outside_range   data_processor.py:78-90
    sample_points                         data_processor.py:145-178
    shuffle_points                        data_processor.py:92-103

The reference runs them per frame in NumPy inside DataLoader workers and uploads the sampled
points afterwards (models/__init__.py:23-34).  Here the RAW frames are uploaded once and one
kernel launch (det6d_prepare_points) masks, samples, shuffles and writes the collated
`points (B*N, 1+C)` tensor of `DatasetTemplate.collate_batch` (datasets/dataset.py:171-176).
Random draws are keyed by (seed, frame id), not by a worker-global generator, so a frame is sampled
identically however it is batched or sharded (include/det6d_rng.h).

The voxel / image / depth-map processors of the reference are outside the hot path and raise here.
"""
import numpy as np
import torch

from ....ops import fused

_SUPPORTED = ('mask_points_and_boxes_outside_range', 'sample_points', 'shuffle_points')


def _get(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


def collate_frames(frames, device='cuda', pinned=None):
    """list of (n_i, C) float32 frames -> (raw (sum n_i, C) device tensor, offsets (B+1) int32 device tensor).
    One pinned staging copy and one asynchronous H2D instead of one upload per key and frame."""
    sizes = [int(f.shape[0]) for f in frames]
    c = int(frames[0].shape[1])
    total = int(sum(sizes))
    offsets = np.zeros(len(frames) + 1, np.int32)
    offsets[1:] = np.cumsum(sizes)
    if pinned is None or pinned.shape[0] < max(total, 1) or pinned.shape[1] != c:
        pinned = torch.empty((max(total, 1), c), dtype=torch.float32)
        if torch.cuda.is_available():
            pinned = pinned.pin_memory()
    view = pinned.numpy()
    for f, lo in zip(frames, offsets[:-1]):
        view[lo:lo + f.shape[0]] = np.asarray(f, np.float32)
    raw = pinned[:max(total, 1)].to(device, non_blocking=True)[:total]
    return raw, torch.from_numpy(offsets).to(device, non_blocking=True), pinned


class DataProcessor(object):
    def __init__(self, processor_configs, point_cloud_range, training, num_point_features=4, seed=0):
        self.point_cloud_range = np.asarray(point_cloud_range, np.float32)
        self.training = training
        self.num_point_features = num_point_features
        self.mode = 'train' if training else 'test'
        self.seed = int(seed)
        self.grid_size = self.voxel_size = None
        self.mask_range = False
        self.num_points = -1
        self.shuffle = False
        self.data_processor_queue = []
        for cur_cfg in processor_configs:
            name = _get(cur_cfg, 'NAME')
            if name not in _SUPPORTED:
                raise NotImplementedError('%s is not on the Det6D point path (supported: %s)' % (name, ', '.join(_SUPPORTED)))
            getattr(self, '_configure_' + name)(cur_cfg)
            self.data_processor_queue.append(name)
        self._pinned = None

    # -- the YAML entries only set switches; the work happens in one fused launch ------------------
    def _configure_mask_points_and_boxes_outside_range(self, cfg):
        self.mask_range = True

    def _configure_sample_points(self, cfg):
        self.num_points = int(_get(_get(cfg, 'NUM_POINTS'), self.mode))

    def _configure_shuffle_points(self, cfg):
        self.shuffle = bool(_get(_get(cfg, 'SHUFFLE_ENABLED'), self.mode))

    def _range_xy(self):
        if self.mask_range:
            return self.point_cloud_range
        big = np.float32(3.0e38)
        return np.asarray([-big, -big, -big, big, big, big], np.float32)

    def forward_batch(self, frames, frame_ids=None, device='cuda'):
        """frames: list of (n_i, C) numpy arrays or ONE (raw, offsets) pair already on the device.
        Returns the batch_dict the detector consumes: points (B*N, 1+C), batch_size, num_in_range."""
        if self.num_points <= 0:
            raise NotImplementedError('sample_points with NUM_POINTS = -1 yields ragged scenes; the batch path needs a fixed count')
        if isinstance(frames, tuple):
            raw, offsets = frames
        else:
            raw, offsets, self._pinned = collate_frames(frames, device, self._pinned)
        b = offsets.numel() - 1
        ids = None
        if frame_ids is not None:
            ids = torch.as_tensor(np.asarray(frame_ids, np.int64) & 0x7fffffff, dtype=torch.int32).to(raw.device, non_blocking=True)
        points, n_in = fused.prepare_points(raw, offsets, self._range_xy(), self.num_points, self.seed, scene_ids=ids)
        return {'points': points, 'batch_size': b, 'num_in_range': n_in}

    def forward(self, data_dict):
        """single-frame form of the reference interface: data_dict['points'] (n, C) -> (N, C) on the device"""
        pts = data_dict['points']
        if torch.is_tensor(pts):
            raw = pts.contiguous().float()
            offsets = torch.tensor([0, raw.shape[0]], dtype=torch.int32, device=raw.device)
            out = self.forward_batch((raw, offsets), frame_ids=[data_dict.get('frame_index', 0)])
        else:
            out = self.forward_batch([np.asarray(pts, np.float32)], frame_ids=[data_dict.get('frame_index', 0)])
        data_dict['points'] = out['points'][:, 1:]
        return data_dict
