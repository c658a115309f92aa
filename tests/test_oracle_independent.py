"""Independent checks of the oracle's restatements of OpenCV behaviour that the reference's own tests do not pin
(no golden vectors; OpenCV itself is absent from /root/reference and from this image).

Every model here is written from the PUBLISHED algorithm (OpenCV 4.x `modules/imgproc/src/drawing.cpp`,
`imgwarp.cpp`, `contours.cpp`) in another form than oracle/*.cpp uses - exact integer / rational arithmetic and
closed forms instead of the library's incremental loops - so that agreement is evidence about the algorithm, not
about a shared transcription:

* cv::fillPoly (postprocess_op.cpp:205,245)  - per-scanline sorted crossings in exact 16.16 integers + closed-form
  Bresenham outline, against oracle_fill_poly (an active-edge-list walk);
* cv::mean with a mask (postprocess_op.cpp:210,251) and the box-threshold decision near 0.4;
* cv::warpPerspective (utility.cpp:178-180) - extended-precision source coordinates + exactly rounded rational
  bilinear value, against the oracle's double / 15-bit fixed-point arithmetic;
* cv::getPerspectiveTransform - exact rational solve of the 8x8 system;
* cv::findContours ordering (postprocess_op.cpp:268-272) - raster discovery order from scipy labelling, reversed,
  and which candidates the `max_candidates = 1000` cut keeps.
"""
import ctypes as C
from fractions import Fraction

import numpy as np
import pytest


def _lib():
    import oracle as O
    return O.lib()


def _p(a):
    return C.c_void_p(a.ctypes.data)


# ------------------------------------------------------------------------------------------------ cv::fillPoly
XY_SHIFT = 16
XY_ONE = 1 << XY_SHIFT


def _cdiv(a, b):
    """C++ integer division (truncation toward zero) on Python ints"""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def _clip_line(w, h, x1, y1, x2, y2):
    """cv::clipLine (Cohen-Sutherland with the library's double arithmetic and int64 truncation)"""
    right, bottom = w - 1, h - 1
    code = lambda x, y: (x < 0) + (x > right) * 2 + (y < 0) * 4 + (y > bottom) * 8
    c1, c2 = code(x1, y1), code(x2, y2)
    if (c1 & c2) == 0 and (c1 | c2) != 0:
        if c1 & 12:
            a = 0 if c1 < 8 else bottom
            x1 += int(float(a - y1) * (x2 - x1) / (y2 - y1))
            y1 = a
            c1 = (x1 < 0) + (x1 > right) * 2
        if c2 & 12:
            a = 0 if c2 < 8 else bottom
            x2 += int(float(a - y2) * (x2 - x1) / (y2 - y1))
            y2 = a
            c2 = (x2 < 0) + (x2 > right) * 2
        if (c1 & c2) == 0 and (c1 | c2) != 0:
            if c1:
                a = 0 if c1 == 1 else right
                y1 += int(float(a - x1) * (y2 - y1) / (x2 - x1))
                x1 = a
                c1 = 0
            if c2:
                a = 0 if c2 == 1 else right
                y2 += int(float(a - x2) * (y2 - y1) / (x2 - x1))
                x2 = a
                c2 = 0
    return (c1 | c2) == 0, x1, y1, x2, y2


def _line_pixels(w, h, p1, p2):
    """The 8-connected cv::line between two points (LineIterator, left to right).  Closed form of its error
    recurrence: along the major axis step i the minor coordinate has moved floor((2*i*dminor + dmajor - 1) / (2*dmajor))
    pixels, i.e. i*dminor/dmajor rounded half DOWN (verified against the recurrence for all dmajor <= 40)."""
    x1, y1, x2, y2 = p1[0], p1[1], p2[0], p2[1]
    if not (0 <= x1 < w and 0 <= x2 < w and 0 <= y1 < h and 0 <= y2 < h):
        ok, x1, y1, x2, y2 = _clip_line(w, h, x1, y1, x2, y2)
        if not ok:
            return []
    if x2 < x1:                                   # the iterator always walks from the left end point
        x1, y1, x2, y2 = x2, y2, x1, y1
    dx, dy = x2 - x1, abs(y2 - y1)
    sy = 1 if y2 >= y1 else -1
    out = []
    if dy > dx:                                   # steep: y is the major axis
        for i in range(dy + 1):
            out.append((x1 + ((2 * i * dx + dy - 1) // (2 * dy) if dy else 0), y1 + sy * i))
    else:
        for i in range(dx + 1):
            out.append((x1 + i, y1 + sy * ((2 * i * dy + dx - 1) // (2 * dx) if dx else 0)))
    return [(x, y) for x, y in out if 0 <= x < w and 0 <= y < h]


CV_45, CV_410 = 45, 410   # include/ocr_hip.h: OCR_CV_45 (OpenCV 4.5.1 rules), OCR_CV_410 (4.5.2 and later; the default)


def fill_poly_model(w, h, pts, cv_compat=CV_410):
    """cv::fillPoly of one polygon with colour 1, line type 8, shift 0 (drawing.cpp).
    CV_45: the classic rule of OpenCV 4.0 - 4.5.1 - edge x in 16.16 fixed point at the vertices' integer x, interior span
    = [ceil(xa), floor(xb)] per scanline.
    CV_410: the rule of 4.5.2 and later - an edge whose outline segment lies inside the image is moved right by half a
    pixel, an edge whose segment had to be clipped runs through the CLIPPED integer end points (no half pixel) and is
    extrapolated back to its first row; the span is [floor(xa), floor(xb)], i.e. both crossings rounded half up when the
    edge was shifted."""
    mask = np.zeros((h, w), np.uint8)
    n = len(pts)
    edges = []
    for i in range(n):
        (ax, ay), (bx, by) = pts[i - 1], pts[i]
        for x, y in _line_pixels(w, h, (ax, ay), (bx, by)):           # the outline
            mask[y, x] = 1
        if ay == by:
            continue
        # end points the edge is built from: (x in 16.16, y)
        c0, c1 = [ax << XY_SHIFT, ay], [bx << XY_SHIFT, by]
        if cv_compat != CV_45:
            if 0 <= ax < w and 0 <= bx < w and 0 <= ay < h and 0 <= by < h:
                c0[0] += XY_ONE >> 1
                c1[0] += XY_ONE >> 1
            else:
                _, x1, y1, x2, y2 = _clip_line(w, h, ax, ay, bx, by)   # the points as far as clipLine moved them
                if y1 != y2:
                    c0, c1 = [x1 << XY_SHIFT, y1], [x2 << XY_SHIFT, y2]
        dxf = _cdiv(c1[0] - c0[0], c1[1] - c0[1])                     # one truncating division per edge, as the library
        if ay < by:
            edges.append((ay, by, c0[0] + (ay - c0[1]) * dxf, dxf))
        else:
            edges.append((by, ay, c1[0] + (by - c1[1]) * dxf, dxf))
    if len(edges) < 2:
        return mask
    ymin, ymax = min(e[0] for e in edges), max(e[1] for e in edges)
    up = XY_ONE - 1 if cv_compat == CV_45 else 0
    for y in range(max(ymin, 0), min(ymax, h)):
        xs = sorted(x + (y - y0) * d for (y0, y1, x, d) in edges if y0 <= y < y1)
        for a, b in zip(xs[0::2], xs[1::2]):
            xa, xb = (a + up) >> XY_SHIFT, b >> XY_SHIFT
            if xa < w and xb >= 0:
                mask[y, max(xa, 0):min(xb, w - 1) + 1] = 1
    return mask


def oracle_fill(w, h, pts, cv_compat=CV_410):
    m = np.zeros((h, w), np.uint8)
    p = np.ascontiguousarray(np.asarray(pts, np.int32).reshape(-1))
    _lib().oracle_fill_poly(_p(m), w, h, _p(p), len(pts), cv_compat)
    return m


def _quads(rs, n, w, h, spill):
    out = []
    for _ in range(n):
        kind = rs.randint(4)
        if kind == 0:                             # rotated rectangle (what minAreaRect / GetMiniBoxes produce)
            cx, cy = rs.uniform(2, w - 2), rs.uniform(2, h - 2)
            a, b = rs.uniform(1, w / 2), rs.uniform(0.5, h / 3)
            t = rs.uniform(0, np.pi)
            u, v = np.array([np.cos(t), np.sin(t)]), np.array([-np.sin(t), np.cos(t)])
            q = [np.array([cx, cy]) + s1 * a * u + s2 * b * v for s1, s2 in ((-1, -1), (1, -1), (1, 1), (-1, 1))]
            q = [(int(p[0]), int(p[1])) for p in q]       # the reference truncates the corners (postprocess_op.cpp:239-242)
        elif kind == 1:                           # arbitrary quad, possibly self-intersecting
            q = [(int(rs.randint(-spill, w + spill)), int(rs.randint(-spill, h + spill))) for _ in range(4)]
        elif kind == 2:                           # slivers / repeated points / horizontal edges
            y = int(rs.randint(0, h))
            q = [(int(rs.randint(0, w)), y), (int(rs.randint(0, w)), y), (int(rs.randint(0, w)), int(rs.randint(0, h))),
                 (int(rs.randint(0, w)), y)]
        else:                                     # polygon with more vertices (PolygonScoreAcc: a contour)
            k = rs.randint(5, 12)
            ang = np.sort(rs.uniform(0, 2 * np.pi, k))
            r = rs.uniform(1, min(w, h) / 2, k)
            q = [(int(w / 2 + r[i] * np.cos(ang[i])), int(h / 2 + r[i] * np.sin(ang[i]))) for i in range(k)]
        if spill == 0:
            q = [(min(max(x, 0), w - 1), min(max(y, 0), h - 1)) for x, y in q]
        out.append(q)
    return out


@pytest.mark.parametrize("cv_compat", [CV_45, CV_410])
@pytest.mark.parametrize("spill", [0, 6])
def test_fill_poly_equals_the_scanline_definition(built, spill, cv_compat):
    """10 000 polygons: rotated rectangles with truncated corners, arbitrary and degenerate quads, contours; with spill > 0
    vertices leave the mask (a box poking out of the probability map: the outline is then drawn between CLIPPED end
    points, which is not the unclipped line restricted to the mask)."""
    rs = np.random.RandomState(100 + spill)
    bad = 0
    for it in range(5000):
        w, h = int(rs.randint(1, 40)), int(rs.randint(1, 30))
        for q in _quads(rs, 1, w, h, spill):
            a, b = fill_poly_model(w, h, q, cv_compat), oracle_fill(w, h, q, cv_compat)
            if not np.array_equal(a, b):
                bad += 1
                if bad <= 3:
                    print("polygon", q, "mask", w, h, "\nmodel\n", a, "\noracle\n", b)
    assert bad == 0


def test_fill_rules_differ_only_at_span_ends(built):
    """The two rules fill the same outline; per scanline the 4.5.2+ span can only gain the pixel a crossing with
    fraction >= 1/2 rounds up to (left end: lose it; right end: gain it) - a sanity check that the switch does what
    DESIGN.md section 5 says, and that it matters (masks differ for a good share of rotated rectangles)."""
    rs = np.random.RandomState(5)
    differ = 0
    for q in _quads(rs, 600, 60, 40, 0):
        a, b = fill_poly_model(60, 40, q, CV_45).astype(int), fill_poly_model(60, 40, q, CV_410).astype(int)
        d = a != b
        differ += bool(d.any())
        assert (d.sum(axis=1) <= len(q)).all()      # at most one pixel per crossing and row
    assert differ > 100


@pytest.mark.parametrize("cv_compat", [CV_45, CV_410])
def test_box_score_is_the_masked_mean_and_agrees_near_the_threshold(built, cv_compat):
    """BoxScoreFast (postprocess_op.cpp:216-253): bounding box from floor / ceil of the corners clamped to the map, mask =
    fillPoly of the truncated corners minus (xmin, ymin), score = cv::mean(pred(roi), mask)[0] (double) returned as
    float.  On maps constructed so that the mean lies within 1e-3 of det_db_box_thresh the oracle's score equals the
    model's float and both take the same side of `score < box_thresh` (postprocess_op.cpp:298)."""
    rs = np.random.RandomState(7)
    L = _lib()
    L.oracle_box_score_fast.restype = C.c_float
    H, W, thr = 60, 80, np.float32(0.4)
    near = 0
    for it in range(3000):
        cx, cy = rs.uniform(5, W - 5), rs.uniform(5, H - 5)
        a, b, t = rs.uniform(2, 30), rs.uniform(1, 12), rs.uniform(0, np.pi)
        u, v = np.array([np.cos(t), np.sin(t)]), np.array([-np.sin(t), np.cos(t)])
        box = np.array([np.array([cx, cy]) + s1 * a * u + s2 * b * v for s1, s2 in ((-1, -1), (1, -1), (1, 1), (-1, 1))], np.float32)
        xs, ys = box[:, 0], box[:, 1]
        clampi = lambda v_, lo, hi: int(min(max(v_, lo), hi))
        xmin, xmax = clampi(np.floor(xs.min()), 0, W - 1), clampi(np.ceil(xs.max()), 0, W - 1)
        ymin, ymax = clampi(np.floor(ys.min()), 0, H - 1), clampi(np.ceil(ys.max()), 0, H - 1)
        pts = [(int(x) - xmin, int(y) - ymin) for x, y in box]
        mask = fill_poly_model(xmax - xmin + 1, ymax - ymin + 1, pts, cv_compat).astype(bool)
        if not mask.any():
            continue
        # a map whose masked mean is thr + d, |d| <= 1e-3, with per-pixel noise that keeps the mean
        d = rs.uniform(-1e-3, 1e-3)
        pred = rs.uniform(0, 1, (H, W)).astype(np.float32)
        roi = pred[ymin:ymax + 1, xmin:xmax + 1]
        noise = rs.uniform(-0.2, 0.2, int(mask.sum()))
        roi[mask] = (float(thr) + d + noise - noise.mean()).astype(np.float32)
        want = np.float32(roi[mask].astype(np.float64).sum() / mask.sum())
        got = np.float32(L.oracle_box_score_fast(_p(np.ascontiguousarray(box.reshape(-1))), _p(pred), H, W, cv_compat))
        assert got == want or abs(float(got) - float(want)) <= 1e-7, (it, got, want)   # (double sum: order-insensitive to 1e-12)
        if abs(float(want) - float(thr)) > 1e-7:
            assert (got < thr) == (want < thr)
        near += abs(float(want) - float(thr)) <= 1.1e-3
    assert near > 2000


# ------------------------------------------------------------------------------------------------ perspective warp
def _exact_perspective(src, dst):
    """the 8 unknowns of cv::getPerspectiveTransform by exact rational Gauss-Jordan"""
    A = [[Fraction(0)] * 9 for _ in range(8)]
    for i in range(4):
        sx, sy, dx, dy = (Fraction(float(v)) for v in (src[i][0], src[i][1], dst[i][0], dst[i][1]))
        A[i][:] = [sx, sy, 1, 0, 0, 0, -sx * dx, -sy * dx, dx]
        A[i + 4][:] = [0, 0, 0, sx, sy, 1, -sx * dy, -sy * dy, dy]
    for c in range(8):
        piv = next(r for r in range(c, 8) if A[r][c] != 0)
        A[c], A[piv] = A[piv], A[c]
        A[c] = [v / A[c][c] for v in A[c]]
        for r in range(8):
            if r != c and A[r][c] != 0:
                A[r] = [a - A[r][c] * b for a, b in zip(A[r], A[c])]
    return [A[r][8] for r in range(8)] + [Fraction(1)]


def _warp_quads(rs, n):
    out = []
    while len(out) < n:
        w, h = rs.randint(8, 60), rs.randint(5, 24)
        q = np.array([[0, 0], [w, 0], [w, h], [0, h]], np.float64)
        t = np.deg2rad(rs.uniform(-35, 35))
        R = np.array([[np.cos(t), np.sin(t)], [-np.sin(t), np.cos(t)]])
        q = (q - q.mean(0)) @ R + rs.uniform(-2.5, 2.5, (4, 2))
        q = np.round(q - q.min(0)).astype(np.int32)
        if np.ptp(q[:, 0]) > 2 and np.ptp(q[:, 1]) > 2:
            out.append(q)
    return out


def test_perspective_transform_equals_the_exact_solution(built):
    rs = np.random.RandomState(3)
    L = _lib()
    for q in _warp_quads(rs, 200):
        cw = int(np.sqrt(float((q[0, 0] - q[1, 0]) ** 2 + (q[0, 1] - q[1, 1]) ** 2)))
        ch = int(np.sqrt(float((q[0, 0] - q[3, 0]) ** 2 + (q[0, 1] - q[3, 1]) ** 2)))
        if cw < 1 or ch < 1:
            continue
        src = q.astype(np.float32)
        dst = np.array([[0, 0], [cw, 0], [cw, ch], [0, ch]], np.float32)
        M = np.zeros(9)
        L.oracle_perspective_transform(_p(np.ascontiguousarray(src.reshape(-1))), _p(np.ascontiguousarray(dst.reshape(-1))), _p(M))
        want = _exact_perspective(src, dst)
        scale = max(abs(float(v)) for v in want)
        assert max(abs(M[i] - float(want[i])) for i in range(9)) <= 1e-9 * max(1.0, scale)


def _warp_model(src, M, dh, dw):
    """cv::warpPerspective(INTER_LINEAR, BORDER_CONSTANT 0) by definition: destination pixel -> M^-1 -> source position
    rounded to 1/32 pixel (INTER_BITS = 5), bilinear taps with weights (32-fy)(32-fx)/1024 etc., value rounded half up.
    Coordinates in extended precision from the EXACT inverse of the double matrix; pixels whose scaled coordinate is
    within 1e-6 of a rounding tie are reported as undecidable (the library's double arithmetic may fall either way)."""
    sh, sw = src.shape[:2]
    Mq = [[Fraction(float(M[3 * r + c])) for c in range(3)] for r in range(3)]
    det = (Mq[0][0] * (Mq[1][1] * Mq[2][2] - Mq[1][2] * Mq[2][1]) - Mq[0][1] * (Mq[1][0] * Mq[2][2] - Mq[1][2] * Mq[2][0])
           + Mq[0][2] * (Mq[1][0] * Mq[2][1] - Mq[1][1] * Mq[2][0]))
    adj = [[Mq[1][1] * Mq[2][2] - Mq[1][2] * Mq[2][1], Mq[0][2] * Mq[2][1] - Mq[0][1] * Mq[2][2], Mq[0][1] * Mq[1][2] - Mq[0][2] * Mq[1][1]],
           [Mq[1][2] * Mq[2][0] - Mq[1][0] * Mq[2][2], Mq[0][0] * Mq[2][2] - Mq[0][2] * Mq[2][0], Mq[0][2] * Mq[1][0] - Mq[0][0] * Mq[1][2]],
           [Mq[1][0] * Mq[2][1] - Mq[1][1] * Mq[2][0], Mq[0][1] * Mq[2][0] - Mq[0][0] * Mq[2][1], Mq[0][0] * Mq[1][1] - Mq[0][1] * Mq[1][0]]]
    inv = np.array([[np.longdouble(float(v / det)) + np.longdouble(float(v / det - Fraction(float(v / det)))) for v in row] for row in adj],
                   dtype=np.longdouble)
    ys, xs = np.mgrid[0:dh, 0:dw]
    xs, ys = xs.astype(np.longdouble), ys.astype(np.longdouble)
    Wd = inv[2, 0] * xs + inv[2, 1] * ys + inv[2, 2]
    X = (inv[0, 0] * xs + inv[0, 1] * ys + inv[0, 2]) / Wd * 32
    Y = (inv[1, 0] * xs + inv[1, 1] * ys + inv[1, 2]) / Wd * 32
    unsure = (np.abs(X - np.floor(X) - 0.5) < 1e-6) | (np.abs(Y - np.floor(Y) - 0.5) < 1e-6)
    Xi, Yi = np.rint(X).astype(np.int64), np.rint(Y).astype(np.int64)
    sx, sy, fx, fy = Xi >> 5, Yi >> 5, Xi & 31, Yi & 31
    out = np.zeros((dh, dw, 3), np.int64)
    pad = np.zeros((sh + 2, sw + 2, 3), np.int64)
    pad[1:-1, 1:-1] = src
    inside = (sx >= -1) & (sx < sw) & (sy >= -1) & (sy < sh)
    cx, cy = np.clip(sx, -1, sw - 1) + 1, np.clip(sy, -1, sh - 1) + 1
    for k in range(3):
        p00, p01 = pad[cy, cx, k], pad[cy, cx + 1, k]
        p10, p11 = pad[cy + 1, cx, k], pad[cy + 1, cx + 1, k]
        num = p00 * (32 - fy) * (32 - fx) + p01 * (32 - fy) * fx + p10 * fy * (32 - fx) + p11 * fy * fx   # / 1024
        out[..., k] = np.where(inside, (2 * num + 1024) // 2048, 0)                                       # round half up
    return out.astype(np.uint8), unsure


def test_warp_perspective_equals_the_rational_bilinear_definition(built):
    rs = np.random.RandomState(5)
    L = _lib()
    src = rs.randint(0, 256, (40, 70, 3)).astype(np.uint8)
    checked = skipped = 0
    for q in _warp_quads(rs, 60):
        cw = int(np.sqrt(float((q[0, 0] - q[1, 0]) ** 2 + (q[0, 1] - q[1, 1]) ** 2)))
        ch = int(np.sqrt(float((q[0, 0] - q[3, 0]) ** 2 + (q[0, 1] - q[3, 1]) ** 2)))
        if cw < 2 or ch < 2:
            continue
        M = np.zeros(9)
        dst = np.array([[0, 0], [cw, 0], [cw, ch], [0, ch]], np.float32)
        L.oracle_perspective_transform(_p(np.ascontiguousarray(q.astype(np.float32).reshape(-1))), _p(np.ascontiguousarray(dst.reshape(-1))), _p(M))
        got = np.zeros((ch, cw, 3), np.uint8)
        L.oracle_warp_perspective(_p(src), src.shape[0], src.shape[1], C.c_size_t(src.strides[0]), _p(M), _p(got), ch, cw)
        want, unsure = _warp_model(src, M, ch, cw)
        ok = (got == want).all(axis=2) | unsure
        assert ok.all(), (q.tolist(), np.argwhere(~ok)[:5].tolist())
        checked += int((~unsure).sum())
        skipped += int(unsure.sum())
    assert checked > 20000 and skipped < checked // 200


# ------------------------------------------------------------------------------------------------ contour order
def _discovery_keys(bm):
    """raster position at which the Suzuki-Abe scan meets each border: an outer border at its component's first pixel in
    raster order, a hole border at the foreground pixel left of the hole's first pixel"""
    from scipy import ndimage
    H, W = bm.shape
    fg, nf = ndimage.label(bm, structure=np.ones((3, 3)))
    keys = {}
    for lab in range(1, nf + 1):
        ys, xs = np.nonzero(fg == lab)
        i = np.lexsort((xs, ys))[0]
        keys[("outer", lab)] = (int(ys[i]), int(xs[i]))
    padded = np.pad(bm == 0, 1, constant_values=True)
    bg, nb = ndimage.label(padded, structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    frame = bg[0, 0]
    for lab in range(1, nb + 1):
        if lab == frame:
            continue
        ys, xs = np.nonzero(bg[1:-1, 1:-1] == lab)
        i = np.lexsort((xs, ys))[0]
        keys[("hole", lab)] = (int(ys[i]), int(xs[i]) - 1)
    return keys


def test_contours_come_in_reverse_discovery_order(built):
    """cv::findContours(RETR_LIST) of the legacy implementation hands the borders back last-found first; the reference
    then keeps the FIRST max_candidates = 1000 of that list (postprocess_op.cpp:271-272), i.e. the 1000 borders the
    raster scan met last."""
    import oracle as O
    rs = np.random.RandomState(2)
    bm = np.zeros((120, 160), np.uint8)
    for _ in range(120):
        y, x, h, w = rs.randint(0, 110), rs.randint(0, 150), rs.randint(1, 10), rs.randint(1, 12)
        bm[y:y + h, x:x + w] = 1
    for _ in range(60):
        y, x = rs.randint(1, 118), rs.randint(1, 158)
        bm[y, x] = 0                                           # holes (and notches)
    bm[0, :] = bm[-1, :] = 0                                   # (the library clears the frame)
    bm[:, 0] = bm[:, -1] = 0
    pts = np.zeros(2 * 200000, np.int32)
    sizes = np.zeros(5000, np.int32)
    n = O.lib().oracle_find_contours(_p(bm), bm.shape[0], bm.shape[1], _p(pts), 200000, _p(sizes), 5000)
    keys = sorted(_discovery_keys(bm).values())
    assert n == len(keys)
    starts = []
    k = 0
    for i in range(n):
        starts.append((int(pts[2 * k + 1]), int(pts[2 * k])))      # a border's first point is where the scan met it
        k += sizes[i]
    assert starts == keys[::-1]


def test_candidate_cut_keeps_the_last_thousand_borders(built):
    import oracle as O
    H, W = 400, 600
    pred = np.full((H, W), 0.02, np.float32)
    cells = [(r, c) for r in range(25) for c in range(44)]       # 1100 blobs, 12 x 12 pitch... 16 x 13
    for r, c in cells:
        pred[r * 16 + 3:r * 16 + 12, c * 13 + 2:c * 13 + 11] = 0.9
    boxes = O.det_post(pred, 0.2, 0.4, 1.8, H, W, cap=2000)
    assert len(boxes) == 1000
    # the raster scan meets blobs row by row; the first 100 (the top rows) are cut
    cy = sorted(float(b[:, 1].mean()) for b in boxes)
    first_kept_row = 100 // 44
    assert cy[0] > first_kept_row * 16, cy[:3]
    kept_cells = {(int(b[:, 1].mean()) // 16, int(b[:, 0].mean()) // 13) for b in boxes}
    assert kept_cells == set(cells[100:])
