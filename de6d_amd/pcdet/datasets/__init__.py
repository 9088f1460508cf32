"""Input side of the hot path: the device-resident data processor."""
from .processor.data_processor import DataProcessor, collate_frames  # noqa: F401
