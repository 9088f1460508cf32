"""The backbone configuration of tests/golden/fp.npz (shared by make_golden.py: gen_fp and the tests): the tiny backbone of
synthetic_models/det6d_tiny.yaml plus two feature-propagation stages (FP_MLPS, pointnet2_backbone.py:178-191)."""
FP_BACKBONE = dict(
    NAME='PointNet2FSMSG',
    SA_CONFIG=dict(NPOINT_LIST=[[512], [128, 128], [64, 64]],
                   SAMPLE_RANGE_LIST=[[[0, 2048]], [[0, 512], [0, 512]], [[0, 128], [128, 256]]],
                   SAMPLE_METHOD_LIST=[['d-fps'], ['s-fps', 'd-fps'], ['s-fps', 'd-fps']],
                   RADIUS=[[0.5, 1.5], [1.5, 3.0], [3.0, 6.0]], NSAMPLE=[[16, 32], [16, 32], [16, 32]],
                   MLPS=[[[8, 8, 16], [8, 16, 16]], [[16, 16, 32], [16, 24, 32]], [[32, 32, 64], [32, 48, 64]]],
                   AGGREGATION_MLPS=[[16], [32], [64]], CONFIDENCE_MLPS=[[8], [16], []], WEIGHT_GAMMA=1.0,
                   DILATED_RADIUS_GROUP=True),
    FP_MLPS=[[40, 24], [48]])
