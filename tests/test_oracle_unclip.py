"""The oracle's ClipperOffset restatement against golden vectors produced by the REFERENCE's own
compiled clipper.cpp (tests/golden/make_fixtures.py), and live against oracle/_ref when present."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_offset(lib, quad, delta):
    xy = np.array(quad, np.int64).ravel()
    out = np.zeros(4000, np.int64)
    lib.oracle_clipper_offset.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_int]
    k = lib.oracle_clipper_offset(xy.ctypes.data, 4, float(delta), out.ctypes.data, 2000)
    return out[:2 * k].reshape(-1, 2).tolist()


def test_unclip_golden_vectors(built):
    import oracle as O
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "unclip_ref.json")))
    assert len(cases) >= 500
    kinds = {"empty": 0, "zero_delta": 0}
    for c in cases:
        got = _oracle_offset(O.lib(), c["quad"], c["delta"])
        assert got == c["out"], (c["quad"], c["delta"])
        kinds["empty"] += c["paths"] == 0
        kinds["zero_delta"] += c["delta"] == 0.0
    assert kinds["empty"] > 0 and kinds["zero_delta"] > 0  # degenerate cases are covered


def test_unclip_live_reference(built):
    import oracle as O
    ref = os.path.join(ROOT, "oracle", "_ref", "libclipper_ref.so")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref not built (reference checkout absent)")
    R = C.CDLL(ref)
    R.clipper_ref_offset.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    rs = np.random.RandomState(99)
    out = np.zeros(4000, np.int64)
    ps = np.zeros(8, np.int32)
    npth = C.c_int()
    slivers = 0
    for t in range(6000):
        cx, cy = rs.rand(2) * 900 + 30
        w, h = rs.rand() * 300 + 1, rs.rand() * 60 + 0.5
        ang = (rs.rand() - 0.5) * math.pi
        ca, sa = math.cos(ang), math.sin(ang)
        quad = [(int(cx + sx * w / 2 * ca - sy * h / 2 * sa), int(cy + sx * w / 2 * sa + sy * h / 2 * ca))
                for sx, sy in ((-1, -1), (1, -1), (1, 1), (-1, 1))]
        d = float(np.float32(w * h * 1.8 / (2 * (w + h))))
        xy = np.array(quad, np.int64).ravel()
        k = R.clipper_ref_offset(xy.ctypes.data, 4, d, out.ctypes.data, 2000, ps.ctypes.data, 8, C.byref(npth))
        if _oracle_offset(O.lib(), quad, d) != out[:2 * k].reshape(-1, 2).tolist():
            # known divergence (DESIGN.md "quirks"): a ~1 px thin quad with a repeated corner and delta < 0.5
            # folds over itself and needs the full Vatti union; measured 4 in 120 000 random quads.
            assert d < 0.5 and len(set(quad)) < 4, (quad, d)
            slivers += 1
    assert slivers <= 2


def _golden20k():
    z = np.load(os.path.join(ROOT, "tests", "golden", "unclip_ref_20k.npz"))
    off = np.concatenate([[0], np.cumsum(z["counts"])])
    return z["quads"], z["deltas"], z["paths"], z["counts"], z["points"], off


def is_sliver(quad8, delta):
    """the one class the restatements do not reproduce: delta < 0.5 on an int-truncated quad at most one
    pixel thin (area / longest side <= 1) - its offset ring folds over itself and only the full Vatti union of
    clipper.cpp:1458-1571 says what is left.  Such a candidate's final box is at most ~2 px high: the reference itself drops
    it at postprocess_op.cpp:357 (rect_height <= 4)."""
    p = np.asarray(quad8, np.float64).reshape(4, 2)
    area = 0.0
    for i in range(4):
        j = (i + 1) % 4
        area += p[i, 0] * p[j, 1] - p[i, 1] * p[j, 0]
    longest = max(float(np.hypot(*(p[i] - p[(i + 1) % 4]))) for i in range(4))
    return delta < 0.5 and (longest == 0 or abs(area) / 2 / longest <= 1.0)


def test_unclip_golden_20k(built):
    """the oracle's ClipperOffset on the 20 000 vectors of the reference's compiled clipper.cpp (ratios 1.8 / 2.0, rotated,
    reversed, duplicate-vertex, degenerate, delta = 0, slivers): point for point, the sliver class excluded by its predicate"""
    import oracle as O
    quads, deltas, paths, counts, points, off = _golden20k()
    assert len(deltas) >= 20000 and (paths == 0).sum() > 100 and (deltas == 0).sum() > 100
    bad = slivers = 0
    for i in range(len(deltas)):
        got = _oracle_offset(O.lib(), quads[i].reshape(4, 2).tolist(), deltas[i])
        want = points[off[i]:off[i + 1]].tolist()
        if got != want:
            if is_sliver(quads[i], deltas[i]):
                slivers += 1
            else:
                bad += 1
    assert bad == 0, bad
    assert slivers <= 10, slivers  # measured: 4 of the 420 cases of the class
