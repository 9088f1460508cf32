from .point_head_box6d_vote import PointHeadBox6DVote

# registry by name, core/pcdet/models/dense_heads/__init__.py:13-25 (Det6D path only)
__all__ = {
    'PointHeadBox6DVote': PointHeadBox6DVote,
}
