from .pointnet2_backbone import PointNet2FSMSG

# registry by name, core/pcdet/models/backbones_3d/__init__.py:7-16 (Det6D path only)
__all__ = {
    'PointNet2FSMSG': PointNet2FSMSG,
}
