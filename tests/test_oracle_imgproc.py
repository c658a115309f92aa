"""Properties and cross-checks of the oracle's OpenCV-semantics restatements (no OpenCV exists in
the container: these pin internal consistency, not OpenCV itself — see oracle/oracle_post.cpp header)."""
import ctypes as C

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st


def _contours(O, bm):
    H, W = bm.shape
    pts = np.zeros(2 * 400000, np.int32)
    sizes = np.zeros(20000, np.int32)
    L = O.lib()
    L.oracle_find_contours.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    n = L.oracle_find_contours(np.ascontiguousarray(bm).ctypes.data, H, W, pts.ctypes.data, 400000, sizes.ctypes.data, 20000)
    out, k = [], 0
    for i in range(n):
        out.append(pts[2 * k:2 * (k + sizes[i])].reshape(-1, 2).copy())
        k += sizes[i]
    return out


def test_contours_known_shapes(built):
    import oracle as O
    bm = np.zeros((8, 10), np.uint8)
    bm[2:5, 3:8] = 1
    assert [c.tolist() for c in _contours(O, bm)] == [[[3, 2], [3, 4], [7, 4], [7, 2]]]
    bm = np.zeros((8, 10), np.uint8)
    bm[1:7, 1:9] = 1
    bm[3:5, 3:6] = 0   # RETR_LIST returns the hole border too, and first (reverse discovery order)
    cs = [c.tolist() for c in _contours(O, bm)]
    assert cs[1] == [[1, 1], [1, 6], [8, 6], [8, 1]] and len(cs) == 2 and len(cs[0]) == 8
    bm = np.zeros((5, 7), np.uint8)
    bm[2, 1:6] = 1     # a 1-px run collapses to two points (dropped by the `size() <= 2` gate)
    assert [c.tolist() for c in _contours(O, bm)] == [[[1, 2], [5, 2]]]
    bm = np.ones((4, 5), np.uint8)  # touches every image border: handled through the 1-px zero frame
    assert [c.tolist() for c in _contours(O, bm)] == [[[0, 0], [0, 3], [4, 3], [4, 0]]]


def test_contour_count_is_components_plus_holes(built):
    from scipy import ndimage
    import oracle as O
    rs = np.random.RandomState(0)
    for _ in range(12):
        H, W = rs.randint(20, 160), rs.randint(20, 160)
        f = ndimage.gaussian_filter(rs.rand(H, W), rs.rand() * 3 + 0.5)
        bm = (f > np.percentile(f, rs.randint(30, 80))).astype(np.uint8)
        cs = _contours(O, bm)
        _, nfg = ndimage.label(bm, structure=np.ones((3, 3)))
        _, nbg = ndimage.label(np.pad(1 - bm, 1, constant_values=1))
        assert len(cs) == nfg + nbg - 1
        assert all(bm[p[:, 1], p[:, 0]].all() for p in cs)


def test_min_area_rect_is_minimal(built):
    from scipy.spatial import ConvexHull
    import oracle as O
    L = O.lib()
    L.oracle_min_area_rect_i.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    rs = np.random.RandomState(0)
    for _ in range(100):
        pts = rs.randint(0, 300, (rs.randint(5, 60), 2)).astype(np.int32)
        out = np.zeros(5, np.float32)
        L.oracle_min_area_rect_i(np.ascontiguousarray(pts).ctypes.data, len(pts), out.ctypes.data)
        h = pts[ConvexHull(pts).vertices].astype(np.float64)
        best = min(np.ptp(h @ e) * np.ptp(h @ np.array([-e[1], e[0]]))
                   for e in ((h[(i + 1) % len(h)] - h[i]) / np.linalg.norm(h[(i + 1) % len(h)] - h[i]) for i in range(len(h))))
        assert abs(out[2] * out[3] - best) / best < 1e-5


def test_resize_identity_area_and_range(built):
    import oracle as O
    rs = np.random.RandomState(0)
    img = rs.randint(0, 256, (40, 60, 3)).astype(np.uint8)
    assert np.array_equal(O.resize_u8c3(img, 40, 60), img)                       # dsize == ssize: copy
    area = ((img[0::2, 0::2].astype(int) + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    assert np.array_equal(O.resize_u8c3(img, 20, 30), area)                      # exact 2x: INTER_AREA
    flat = np.full((17, 23, 3), 131, np.uint8)
    assert (O.resize_u8c3(flat, 48, 91) == 131).all()                            # constants stay constant
    up = O.resize_u8c3(img, 77, 101)
    assert up.min() >= img.min() and up.max() <= img.max()
    view = img[5:25, 7:37]                                                        # ROI view with a row stride
    assert np.array_equal(O.resize_u8c3(view, 48, 100), O.resize_u8c3(view.copy(), 48, 100))


@settings(max_examples=200, deadline=None)
@given(st.integers(1, 3000), st.integers(1, 3000), st.sampled_from(["max", "min"]), st.sampled_from([512, 736, 960]))
def test_det_resize_shape_properties(h, w, limit_type, side):
    import oracle as O
    rh, rw, fh, fw = O.det_resize_shape(h, w, limit_type, side)
    assert rh % 32 == 0 and rw % 32 == 0 and rh >= 32 and rw >= 32
    assert fh == np.float32(rh) / np.float32(h) and fw == np.float32(rw) / np.float32(w)
    if limit_type == "max":
        assert max(rh, rw) <= side + 16 or max(h, w) <= side


@settings(max_examples=200, deadline=None)
@given(st.floats(0, 1, width=32), st.sampled_from([0.2, 0.3, 0.5]))
def test_threshold_is_floor_on_truncated_byte(p, thr):
    """cbuf = (uchar)(p*255); cv::threshold floors thr*255: bit = trunc(p*255) > floor(thr*255)."""
    import oracle as O
    bm = O.bitmap(np.array([[p]], np.float32), thr)
    assert int(bm[0, 0]) == int(int(np.float32(p) * np.float32(255)) > int(np.floor(thr * 255)))


@settings(max_examples=200, deadline=None)
@given(st.lists(st.integers(0, 5), min_size=1, max_size=40))
def test_ctc_collapse_rule(amax):
    import oracle as O
    pmax = np.linspace(0.5, 0.9, len(amax)).astype(np.float32)
    ids, score = O.ctc_decode(np.array(amax, np.int32), pmax)
    want = [a for i, a in enumerate(amax) if a > 0 and not (i > 0 and a == amax[i - 1])]
    if not want:
        assert ids is None        # NaN score: the reference `continue`s and leaves ""/0.0
    else:
        assert ids.tolist() == want
        keep = [pmax[i] for i, a in enumerate(amax) if a > 0 and not (i > 0 and a == amax[i - 1])]
        s = np.float32(0)
        for v in keep:
            s = np.float32(s + v)
        assert score == np.float32(s / np.float32(len(keep)))


def test_crop_rect(built):
    import oracle as O
    assert O.crop_rect([[10, 5], [50, 5], [50, 20], [10, 20]], 100, 100) == (10, 5, 41, 16)
    assert O.crop_rect([[90, 90], [120, 90], [120, 130], [90, 130]], 100, 100) == (90, 90, 10, 10)
    assert O.crop_rect([[100, 0], [120, 0], [120, 5], [100, 5]], 100, 100) is None


def test_rec_preprocess_pad_and_cls_pad(built):
    import oracle as O
    crop = np.random.RandomState(0).randint(0, 256, (20, 30, 3)).astype(np.uint8)
    x = O.rec_preprocess(crop, 48, 320)
    assert x.shape == (48, 320, 3) and (x[:, 72:] == -1.0).all()      # u8 zero pad BEFORE normalise -> -1
    c = O.cls_preprocess(crop)
    assert c.shape == (48, 192, 3) and (c[:, 72:] == 0.0).all()       # cls pads with 0.0 AFTER normalise
    assert np.array_equal(x[:, :72], c[:, :72])


def test_rotate_crop_restatement_properties(built):
    """GetRotateCropImage (utility.cpp:137-190): what must hold whatever the homography arithmetic is."""
    import oracle as O
    rs = np.random.RandomState(4)
    img = rs.randint(0, 256, (150, 220, 3)).astype(np.uint8)
    # axis-aligned box: the homography is the identity, so the warp is the [top,bottom) x [left,right) crop
    assert np.array_equal(O.rotate_crop(img, [[20, 30], [100, 30], [100, 60], [20, 60]]), img[30:60, 20:100])
    # rows >= 1.5 cols: transpose + flip(0) = 90 degrees counter-clockwise
    assert np.array_equal(O.rotate_crop(img, [[20, 30], [40, 30], [40, 100], [20, 100]]), np.rot90(img[30:100, 20:40]))
    assert O.rotate_crop(img, [[20, 30], [40, 30], [40, 59], [20, 59]]).shape == (29, 20, 3)      # 29 < 30: not turned
    assert O.rotate_crop(img, [[20, 30], [40, 30], [40, 60], [20, 60]]).shape == (20, 30, 3)      # 30 >= 30: turned
    # corners listed from another start: the crop turns with them (box given as TR,BR,BL,TL -> 90 degrees cw,
    # then the tall result is turned back by the 1.5 rule)
    q = O.rotate_crop(img, [[100, 30], [100, 60], [20, 60], [20, 30]])
    assert q.shape == (30, 80, 3)
    # a smooth ramp stays a ramp under a rotated quad: bilinear taps, no border leak inside the quad
    yy, xx = np.mgrid[0:150, 0:220]
    ramp = np.stack([xx, yy, (xx + yy) // 2], -1).astype(np.uint8)
    c = O.rotate_crop(ramp, [[20, 40], [100, 30], [104, 60], [24, 70]])
    assert c.shape == (30, 80, 3)
    assert tuple(c[0, 0]) == (20, 40, 30) and abs(int(c[0, -1, 0]) - 99) <= 1 and abs(int(c[-1, 0, 1]) - 69) <= 1
    assert (np.diff(c[:, :, 0].astype(int), axis=1) >= 0).all() and (np.diff(c[:, :, 1].astype(int), axis=0) >= 0).all()
    # the warp reads the CROP, not the image: outside the bounding box taps are 0 (BORDER_CONSTANT; the
    # reference's cv::BORDER_REPLICATE sits in the flags slot, utility.cpp:178-180)
    white = np.full((100, 100, 3), 255, np.uint8)
    c = O.rotate_crop(white, [[10, 10], [60, 10], [70, 40], [10, 40]])     # p2 sticks out: dst corner maps inside
    assert c.min() == 255 or (c == 255).mean() > 0.95
    # empty / outside boxes have no crop
    assert O.rotate_crop(img, [[5, 5], [5, 5], [5, 5], [5, 5]]) is None
    assert O.rotate_crop(img, [[-1, 5], [30, 5], [30, 20], [-1, 20]]) is None


def _expand_simple(poly):
    """CHAIN_APPROX_SIMPLE keeps only the end points of straight runs (8 directions): the full pixel chain back."""
    pts = set()
    n = len(poly)
    for i in range(n):
        (x0, y0), (x1, y1) = poly[i], poly[(i + 1) % n]
        dx, dy = int(np.sign(x1 - x0)), int(np.sign(y1 - y0))
        steps = max(abs(x1 - x0), abs(y1 - y0))
        assert abs(x1 - x0) in (0, steps) and abs(y1 - y0) in (0, steps)   # straight in one of the 8 directions
        for k in range(steps + 1):
            pts.add((x0 + k * dx, y0 + k * dy))
    return pts


def test_contour_point_sets_equal_the_border_definition(built):
    """Second, independently written check of findContours(RETR_LIST, CHAIN_APPROX_SIMPLE) - point SETS, not counts.
    Suzuki-Abe's border between an 8-connected 1-component S1 and a 4-connected 0-component S2 is, by definition, the
    set of pixels of S1 with a 4-neighbour in S2.  Computed here with scipy labelling (no border following at all) and
    compared with the pixel chains the oracle's contours expand to: every (S1, S2) adjacency is one contour, every
    contour is exactly one such set."""
    from scipy import ndimage
    import oracle as O
    rs = np.random.RandomState(7)
    for trial in range(14):
        H, W = rs.randint(16, 120), rs.randint(16, 120)
        f = ndimage.gaussian_filter(rs.rand(H, W), rs.rand() * 2.5 + 0.4)
        bm = (f > np.percentile(f, rs.randint(25, 80))).astype(np.uint8)
        if trial % 4 == 0:
            bm[rs.randint(0, H), :] = 1          # 1-px lines, frame contact
            bm[:, rs.randint(0, W)] = 1
        fg, nfg = ndimage.label(bm, structure=np.ones((3, 3)))
        bgp, nbg = ndimage.label(np.pad(1 - bm, 1, constant_values=1))      # 4-connected zeros, with the outside frame
        want = {}
        ys, xs = np.nonzero(bm)
        for y, x in zip(ys, xs):
            for dy, dx in ((-1, 0), (1, 0), (0, -1), (0, 1)):
                b = bgp[y + 1 + dy, x + 1 + dx]
                if b:
                    want.setdefault((fg[y, x], b), set()).add((x, y))
        got = [_expand_simple([tuple(p) for p in c.tolist()]) for c in _contours(O, bm)]
        assert len(got) == len(want) == nfg + nbg - 1
        left = dict(want)
        for pts in got:
            keys = [k for k, v in left.items() if v == pts]
            assert len(keys) >= 1, "a contour that is no (component, background) border set"
            del left[keys[0]]
        assert not left


def test_fixed_point_resize_against_exact_rational_bilinear(built):
    """cv::resize INTER_LINEAR on 8-bit data works in fixed point (11-bit coefficients, 22-bit products).  An
    independent evaluation: the same sample positions (float32 arithmetic on the coordinates, as OpenCV) but the
    interpolation itself in exact rational arithmetic.  The fixed-point result must be within 1 of the exactly
    rounded value everywhere and equal to it at more than 4 of 5 positions; plus a plain-Python big-int restatement of the
    fixed-point formula (written from the formula, not from the oracle's C) that must match bit for bit."""
    from fractions import Fraction
    import oracle as O
    rs = np.random.RandomState(3)
    for (sh, sw, dh, dw) in [(9, 13, 17, 31), (20, 24, 11, 7), (6, 40, 48, 57), (31, 5, 12, 23), (8, 8, 9, 8)]:
        img = rs.randint(0, 256, (sh, sw, 3)).astype(np.uint8)
        got = O.resize_u8c3(img, dh, dw)
        if dh * 2 == sh and dw * 2 == sw:
            continue
        sx_, sy_ = 1.0 / (np.float64(dw) / sw), 1.0 / (np.float64(dh) / sh)   # resize.cpp: scale = 1 / inv_scale

        def coords(n_dst, n_src, scale, clamp):
            # x: positions left of the first / right of the last pixel centre take that pixel alone (weight 0 on the
            # neighbour); y: the fraction is kept and the two ROWS are clipped instead (resize.cpp, resizeGeneric_)
            out = []
            for d in range(n_dst):
                f = np.float32((d + 0.5) * scale - 0.5)
                s = int(np.floor(f))
                f = np.float32(f - np.float32(s))
                if clamp and s < 0:
                    s, f = 0, np.float32(0)
                if clamp and s >= n_src - 1:
                    s, f = n_src - 1, np.float32(0)
                out.append((s, f))
            return out
        cx, cy = coords(dw, sw, sx_, True), coords(dh, sh, sy_, False)
        sat = lambda v: int(max(-32768, min(32767, int(np.rint(np.float32(v))))))
        worst, exact_hits, total = 0, 0, 0
        for y, (sy, fy) in enumerate(cy):
            for x, (sx, fx) in enumerate(cx):
                x1 = min(sx + 1, sw - 1)
                y0, y1 = min(max(sy, 0), sh - 1), min(max(sy + 1, 0), sh - 1)
                a0, a1 = sat((np.float32(1) - fx) * np.float32(2048)), sat(fx * np.float32(2048))
                b0, b1 = sat((np.float32(1) - fy) * np.float32(2048)), sat(fy * np.float32(2048))
                for c in range(3):
                    p00, p01, p10, p11 = (int(img[y0, sx, c]), int(img[y0, x1, c]), int(img[y1, sx, c]), int(img[y1, x1, c]))
                    # plain big-int restatement of the fixed-point formula
                    r0, r1 = p00 * a0 + p01 * a1, p10 * a0 + p11 * a1
                    v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2
                    assert got[y, x, c] == max(0, min(255, v))
                    # exact rational bilinear at the same float32 sample offsets
                    FX, FY = Fraction(float(fx)), Fraction(float(fy))
                    e = (p00 * (1 - FX) + p01 * FX) * (1 - FY) + (p10 * (1 - FX) + p11 * FX) * FY
                    r = int(e + Fraction(1, 2))
                    worst = max(worst, abs(int(got[y, x, c]) - r))
                    exact_hits += int(got[y, x, c]) == r
                    total += 1
        assert worst <= 1 and exact_hits / total > 0.8, (worst, exact_hits / total)
