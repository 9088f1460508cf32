"""The separated-pair rule of the NMS mask kernels (csrc/iou3d_nms.hip: d6_nms_radius / d6_nms_near) against the REFERENCE's own
iou3d_cpu.cpp (oracle/_ref): a pair the rule skips must never have an IoU above 0 in the reference's arithmetic — so skipping it
cannot change a suppression bit (`iou > thresh`, thresh >= 0).  Random scenes and the adversarial families the argument in the
kernel's comment worries about: rows of collinear axis-aligned boxes, rows sharing a rotated heading, huge coordinates, tiny and
degenerate boxes, pairs right at the rule's boundary."""
import numpy as np
import pytest

from oracle import ref as oref

pytestmark = pytest.mark.skipif(not oref.available(), reason="oracle/_ref (the reference's iou3d_cpu.cpp) is not built here")

f32 = np.float32


def nms_radius(b):
    """d6_nms_radius, operation for operation in fp32"""
    x, y, dx, dy = b[:, 0], b[:, 1], b[:, 3], b[:, 4]
    return (f32(0.5) * np.sqrt(dx * dx + dy * dy) * f32(1.0001) + f32(0.05)) + f32(1e-5) * (np.abs(x) + np.abs(y))


def nms_near(a, b):
    """d6_nms_near for every pair of two sets -> (Na, Nb) bool"""
    ra, rb = nms_radius(a), nms_radius(b)
    ex = a[:, None, 0] - b[None, :, 0]
    ey = a[:, None, 1] - b[None, :, 1]
    s = ra[:, None] + rb[None, :]
    with np.errstate(invalid='ignore', over='ignore'):
        return ~(ex * ex + ey * ey > s * s)


def boxes(rng, k, spread, lo=0.3, hi=8.0, heading=None, z=True):
    b = np.zeros((k, 7), f32)
    b[:, 0] = rng.uniform(-spread, spread, k)
    b[:, 1] = rng.uniform(-spread, spread, k)
    b[:, 2] = rng.uniform(-2, 2, k) if z else 0
    b[:, 3] = rng.uniform(lo, hi, k)
    b[:, 4] = rng.uniform(lo, hi, k)
    b[:, 5] = rng.uniform(1, 3, k)
    b[:, 6] = rng.uniform(-7, 7, k) if heading is None else heading
    return b


def families(rng):
    yield "random scene", boxes(rng, 900, 60.0), boxes(rng, 900, 60.0)
    yield "dense scene", boxes(rng, 900, 12.0), boxes(rng, 900, 12.0)
    # rows of equal axis-aligned boxes: top / bottom edges exactly collinear, spacing from touching to far
    row = boxes(rng, 800, 0.0, heading=0.0)
    row[:, 3], row[:, 4] = f32(3.9), f32(1.6)
    row[:, 0] = (np.arange(800) * f32(0.37)).astype(f32)
    row[:, 1] = f32(2.25)
    yield "collinear axis-aligned row", row, row.copy()
    col = row.copy(); col[:, [0, 1]] = col[:, [1, 0]]; col[:, 6] = f32(np.pi / 2)
    yield "collinear column, right angle", col, col.copy()
    # the same row rotated about the origin by arbitrary headings: edges collinear up to rounding
    for ang in (0.3, 1.1, -2.2, np.pi / 4):
        r = row.copy()
        c, s = np.cos(ang), np.sin(ang)
        r[:, 0], r[:, 1] = (row[:, 0] * c - row[:, 1] * s).astype(f32), (row[:, 0] * s + row[:, 1] * c).astype(f32)
        r[:, 6] = f32(ang)
        yield "rotated row %.2f" % ang, r, r.copy()
    # a lattice: collinear in both directions, mixed 0 / 90 degree headings
    g = boxes(rng, 900, 0.0, heading=0.0)
    gx, gy = np.meshgrid(np.arange(30), np.arange(30))
    g[:, 0], g[:, 1] = (gx.ravel() * f32(2.5)).astype(f32), (gy.ravel() * f32(2.5)).astype(f32)
    g[:, 3], g[:, 4] = f32(2.0), f32(1.0)
    g[::3, 6] = f32(np.pi / 2)
    yield "lattice", g, g.copy()
    far = boxes(rng, 700, 5.0e4)
    yield "huge coordinates", far, far.copy()
    tiny = boxes(rng, 700, 3.0, lo=1e-4, hi=0.05)
    yield "tiny boxes", tiny, boxes(rng, 700, 3.0)
    deg = boxes(rng, 700, 20.0)
    deg[::4, 3] = 0.0; deg[1::4, 4] = 0.0; deg[2::4, 3] = -deg[2::4, 3]
    yield "zero / negative extents", deg, boxes(rng, 700, 20.0)
    # pairs placed right at the rule's boundary: centre distance = (ra + rb) * (1 +- small)
    a = boxes(rng, 900, 30.0)
    bb = boxes(rng, 900, 30.0)
    d = (nms_radius(a) + nms_radius(bb)) * rng.uniform(0.97, 1.03, 900).astype(f32)
    phi = rng.uniform(0, 2 * np.pi, 900)
    bb[:, 0], bb[:, 1] = (a[:, 0] + d * np.cos(phi)).astype(f32), (a[:, 1] + d * np.sin(phi)).astype(f32)
    yield "boundary pairs", a, bb


def test_skipped_pairs_have_no_overlap_in_the_reference():
    rng = np.random.default_rng(20261003)
    skipped = evaluated = 0
    for rep in range(3):
        for name, a, b in families(rng):
            iou = oref.boxes_iou_bev_cpu(a, b)
            near = nms_near(a, b)
            with np.errstate(invalid='ignore'):
                bad = (~near) & (iou > 0)                  # NaN > 0 is False: what `iou > thresh` gives in the mask kernel too
            assert not bad.any(), "%s: the rule skips a pair with IoU %r" % (name, iou[bad][:5])
            skipped += int((~near).sum())
            evaluated += int(near.sum())
    # the rule is worth having: most pairs are skipped, and the test saw millions of them
    assert skipped > 5_000_000 and evaluated > 100_000


def test_rule_never_skips_on_non_finite_input():
    a = np.array([[np.nan, 0, 0, 2, 1, 1, 0], [np.inf, 0, 0, 2, 1, 1, 0], [0, 0, 0, np.nan, 1, 1, 0], [0, 0, 0, np.inf, 1, 1, 0]], f32)
    b = np.array([[500, 500, 0, 2, 1, 1, 0.3]], f32)
    assert nms_near(a, b).all() and nms_near(b, a).all()
