"""Generates the committed fixtures of tests/golden/ (run in the dev container, where
/root/reference exists).

  card_jd_bgr.npy     the reference's only test image (/root/reference/images/card-jd.jpg, used by
                      /root/reference/tests/test_ocr_worker.cpp:182-233), decoded with PIL/libjpeg to
                      BGR u8 [178,391,3].  (OpenCV's own JPEG decode may differ by +-1 LSB: SURVEY §8c G3.)
  unclip_ref_20k.npz  the same call on 20 000 seeded quads (ratios 1.8 and 2.0, rotated, reversed, duplicate-vertex, fully
                      degenerate, delta = 0, the delta < 0.5 sliver class), packed: quads int32 [n,8], deltas f64 [n],
                      paths int8 [n], counts int32 [n], points int32 [sum(counts),2] - vectors only, as unclip_ref.json
  unclip_ref.json     golden vectors of ClipperOffset(jtRound, etClosedPolygon).Execute produced by the
                      REFERENCE's compiled src/clipper.cpp (oracle/_ref/libclipper_ref.so): the exact
                      call DBPostProcessor::UnClip makes (postprocess_op.cpp:46-55).
"""
import ctypes as C
import json
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def card():
    from PIL import Image
    im = np.array(Image.open("/root/reference/images/card-jd.jpg").convert("RGB"))[:, :, ::-1].copy()
    np.save(os.path.join(HERE, "card_jd_bgr.npy"), im)


def unclip(n=600):
    R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libclipper_ref.so"))
    R.clipper_ref_offset.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    rs = np.random.RandomState(20251003)
    out = np.zeros(4000, np.int64)
    ps = np.zeros(8, np.int32)
    npth = C.c_int()
    cases = []
    for t in range(n):
        cx, cy = rs.rand(2) * 900 + 30
        w = rs.rand() * 300 + 1
        h = rs.rand() * 60 + 0.5
        ang = (rs.rand() - 0.5) * (math.pi if t % 3 == 0 else 0.2)
        ca, sa = math.cos(ang), math.sin(ang)
        pts = [(int(cx + sx * w / 2 * ca - sy * h / 2 * sa), int(cy + sx * w / 2 * sa + sy * h / 2 * ca))
               for sx, sy in ((-1, -1), (1, -1), (1, 1), (-1, 1))]
        if t % 7 == 0:
            pts = pts[::-1]
        if t % 41 == 0:
            pts[1] = pts[0]  # duplicate vertex
        if t % 53 == 0:
            pts = [pts[0]] * 4  # fully degenerate
        d = float(np.float32(w * h * (1.8 if t % 2 else 2.0) / (2 * (w + h))))
        if t % 29 == 0:
            d = 0.0
        xy = np.array(pts, np.int64).ravel()
        k = R.clipper_ref_offset(xy.ctypes.data, 4, d, out.ctypes.data, 2000, ps.ctypes.data, 8, C.byref(npth))
        assert k >= 0 and npth.value <= 1
        cases.append(dict(quad=[list(p) for p in pts], delta=d, paths=npth.value,
                          out=out[:2 * k].reshape(-1, 2).tolist()))
    json.dump(cases, open(os.path.join(HERE, "unclip_ref.json"), "w"), separators=(",", ":"))


def unclip20k(n=20000):
    R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libclipper_ref.so"))
    R.clipper_ref_offset.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    rs = np.random.RandomState(20261005)
    out = np.zeros(4000, np.int64)
    ps = np.zeros(8, np.int32)
    npth = C.c_int()
    quads, deltas, paths, counts, points = [], [], [], [], []
    for t in range(n):
        cx, cy = rs.rand(2) * 900 + 30
        w = rs.rand() * 300 + 1
        h = rs.rand() * 60 + 0.5
        if t % 11 == 0:  # the sliver class: ~1 px thin, delta < 0.5
            h = rs.rand() * 1.2 + 0.3
        ang = (rs.rand() - 0.5) * (math.pi if t % 3 == 0 else 0.2)
        ca, sa = math.cos(ang), math.sin(ang)
        pts = [(int(cx + sx * w / 2 * ca - sy * h / 2 * sa), int(cy + sx * w / 2 * sa + sy * h / 2 * ca))
               for sx, sy in ((-1, -1), (1, -1), (1, 1), (-1, 1))]
        if t % 7 == 0:
            pts = pts[::-1]
        if t % 41 == 0:
            pts[1] = pts[0]  # duplicate vertex
        if t % 43 == 0:
            pts[3] = pts[0]  # closing duplicate
        if t % 53 == 0:
            pts = [pts[0]] * 4  # fully degenerate
        d = float(np.float32(w * h * (1.8 if t % 2 else 2.0) / (2 * (w + h))))
        if t % 29 == 0:
            d = 0.0
        xy = np.array(pts, np.int64).ravel()
        k = R.clipper_ref_offset(xy.ctypes.data, 4, d, out.ctypes.data, 2000, ps.ctypes.data, 8, C.byref(npth))
        assert k >= 0 and npth.value <= 1
        quads.append(xy.astype(np.int32)); deltas.append(d); paths.append(npth.value); counts.append(k)
        points.append(out[:2 * k].reshape(-1, 2).astype(np.int32))
    np.savez_compressed(os.path.join(HERE, "unclip_ref_20k.npz"), quads=np.array(quads, np.int32), deltas=np.array(deltas, np.float64),
                        paths=np.array(paths, np.int8), counts=np.array(counts, np.int32), points=np.concatenate(points).astype(np.int32))


if __name__ == "__main__":
    import sys
    if "--only-20k" not in sys.argv:
        card()
        unclip()
    unclip20k()
