"""Input side of the hot path: the device-resident data processor."""
from .processor.data_processor import DataProcessor, collate_frames  # noqa: F401
from .kitti.kitti_dataset import KittiDataset  # noqa: F401,E402
from .slopedkitti.kitti_dataset import SlopedKittiDataset  # noqa: F401,E402

#: DATA_CONFIG.DATASET -> class, the names the reference registers (core/pcdet/datasets/__init__.py:14-20)
__all__ = {'KittiDataset': KittiDataset, 'SlopedKittiDataSet': SlopedKittiDataset}
