"""The oracle against the REFERENCE's own iou3d_cpu.cpp compiled into oracle/_ref (when the
prebuilt library is present: it is built only where /root/reference exists)."""
import numpy as np
import pytest

from oracle import ref as oref
from tests.util import random_boxes

pytestmark = pytest.mark.skipif(not oref.available(), reason="oracle/_ref not built (no /root/reference)")


@pytest.mark.parametrize("seed,k,spread", [(1, 300, 30.0), (2, 128, 6.0), (3, 64, 2.0)])
def test_iou_matrix_and_keep_lists(oracle_ops, seed, k, spread):
    boxes = random_boxes(seed, k, spread=spread)
    ours = oracle_ops.boxes_iou_bev(boxes, boxes)
    theirs = oref.boxes_iou_bev_cpu(boxes, boxes)
    np.testing.assert_allclose(ours, theirs, rtol=0, atol=2e-5)
    for thr in (0.01, 0.1, 0.5):
        np.testing.assert_array_equal(oracle_ops.nms(boxes, thr), oracle_ops.nms_from_iou(theirs, thr))


def test_degenerate_boxes(oracle_ops):
    boxes = random_boxes(9, 40, spread=5.0)
    boxes[1] = boxes[0]
    boxes[2, 3:5] = 0
    boxes[3, 6] = 0.0
    boxes[4, 6] = np.pi / 2
    boxes[5] = boxes[3]; boxes[5, 0] += boxes[3, 3]
    ours = oracle_ops.boxes_iou_bev(boxes, boxes)
    theirs = oref.boxes_iou_bev_cpu(boxes, boxes)
    np.testing.assert_allclose(ours, theirs, rtol=0, atol=2e-5)
