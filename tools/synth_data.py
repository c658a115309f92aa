"""Seeded synthetic workload of BASELINE.json configs[1..3] (SURVEY.md §8d "cfg2"/"cfg3"):
text-card images with K rotated text-line rectangles, and the matching synthetic DB probability
map (used for post-processing + rec when det/rec weights are synthetic, so that box counts and rec
batch shapes are controlled and reproducible).  Pure integer/float64 splitmix64 arithmetic.
"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from synth_weights import _splitmix  # noqa: E402


class Rng:
    def __init__(self, seed):
        self.seed = seed
        self.stream = 0

    def uniform(self, n=1):
        self.stream += 1
        return _splitmix(self.seed, n, self.stream)


def make_layout(seed, H=960, W=960, K=32, wr=(80, 400), hr=(16, 48), max_angle_deg=5.0):
    """-> list of (cx, cy, w, h, angle_rad), non-overlapping, fully inside the image."""
    rng = Rng(seed)
    rects = []
    tries = 0
    while len(rects) < K and tries < 20000:
        tries += 1
        u = rng.uniform(5)
        w = wr[0] + (wr[1] - wr[0]) * u[0]
        h = hr[0] + (hr[1] - hr[0]) * u[1]
        ang = math.radians((2 * u[2] - 1) * max_angle_deg)
        ex = abs(w / 2 * math.cos(ang)) + abs(h / 2 * math.sin(ang)) + 6
        ey = abs(w / 2 * math.sin(ang)) + abs(h / 2 * math.cos(ang)) + 6
        if 2 * ex >= W or 2 * ey >= H:
            continue
        cx = ex + (W - 2 * ex) * u[3]
        cy = ey + (H - 2 * ey) * u[4]
        ok = True
        for (ox, oy, oex, oey) in [(r[0], r[1], r[5], r[6]) for r in rects]:
            if abs(cx - ox) < ex + oex + 4 and abs(cy - oy) < ey + oey + 4:
                ok = False
                break
        if ok:
            rects.append((cx, cy, w, h, ang, ex, ey))
    return [r[:5] for r in rects]


def _local(H, W, r):
    cx, cy, w, h, ang = r
    ex = abs(w / 2 * math.cos(ang)) + abs(h / 2 * math.sin(ang)) + 4
    ey = abs(w / 2 * math.sin(ang)) + abs(h / 2 * math.cos(ang)) + 4
    x0, x1 = max(0, int(cx - ex)), min(W, int(cx + ex) + 1)
    y0, y1 = max(0, int(cy - ey)), min(H, int(cy + ey) + 1)
    yy, xx = np.mgrid[y0:y1, x0:x1]
    dx, dy = xx - cx, yy - cy
    u = dx * math.cos(ang) + dy * math.sin(ang)
    v = -dx * math.sin(ang) + dy * math.cos(ang)
    return (y0, y1, x0, x1), u, v


def make_image(seed, H=960, W=960, K=32, layout=None):
    """BGR u8 image [H,W,3]: background 230+-8 noise, dark vertical strokes inside each rectangle."""
    layout = layout if layout is not None else make_layout(seed, H, W, K)
    rng = Rng(seed ^ 0x5EED1)
    img = (222 + 16 * rng.uniform(H * W * 3)).reshape(H, W, 3)
    for i, r in enumerate(layout):
        (y0, y1, x0, x1), u, v = _local(H, W, r)
        w, h = r[2], r[3]
        period = 6 + 8 * rng.uniform(1)[0]
        inside = (np.abs(u) <= w / 2) & (np.abs(v) <= h / 2 * 0.8)
        stroke = inside & (((u + w / 2) % period) < period / 2)
        dark = (20 + 40 * rng.uniform(stroke.size)).reshape(stroke.shape)
        sub = img[y0:y1, x0:x1]
        for c in range(3):
            sub[..., c] = np.where(stroke, dark, sub[..., c])
    return np.clip(img, 0, 255).astype(np.uint8)


def make_prob_map(seed, H=960, W=960, K=32, layout=None):
    """f32 [H,W]: 0.9 inside each rectangle, 2-px linear ramp, 0.02 outside, + U[0,0.01]."""
    layout = layout if layout is not None else make_layout(seed, H, W, K)
    rng = Rng(seed ^ 0xB17)
    p = np.full((H, W), 0.02, np.float64)
    for r in layout:
        (y0, y1, x0, x1), u, v = _local(H, W, r)
        d = np.maximum(np.abs(u) - r[2] / 2, np.abs(v) - r[3] / 2)
        q = np.where(d <= 0, 0.9, np.where(d < 2, 0.9 - 0.88 * d / 2, 0.02))
        p[y0:y1, x0:x1] = np.maximum(p[y0:y1, x0:x1], q)
    p += 0.01 * rng.uniform(H * W).reshape(H, W)
    return p.astype(np.float32)


def cfg2_sample(i, H=960, W=960, K=32):
    seed = 1000 + i
    layout = make_layout(seed, H, W, K)
    return make_image(seed, H, W, K, layout), make_prob_map(seed, H, W, K, layout), layout


def cfg3_prob_at(i, rh, rw):
    """The synthetic probability map of cfg3 sample i at the detector's input resolution rh x rw (the layout's
    rectangles scaled per axis; angles are within 5 degrees, so a scaled rectangle stays a rectangle to a pixel)."""
    seed = 2000 + i
    u = Rng(seed ^ 0xC3).uniform(3)
    H = 640 + int(u[0] * 641)
    W = 640 + int(u[1] * 641)
    K = 4 + int(u[2] * 61)
    sx, sy = rw / W, rh / H
    layout = [(cx * sx, cy * sy, w * sx, h * sy, ang) for (cx, cy, w, h, ang) in make_layout(seed, H, W, K)]
    return make_prob_map(seed, rh, rw, K, layout)


def cfg3_sample(i):
    """mixed-aspect 640-1280 px images, K in 4..64 (cls on)."""
    seed = 2000 + i
    u = Rng(seed ^ 0xC3).uniform(3)
    H = 640 + int(u[0] * 641)
    W = 640 + int(u[1] * 641)
    K = 4 + int(u[2] * 61)
    layout = make_layout(seed, H, W, K)
    return make_image(seed, H, W, K, layout), make_prob_map(seed, H, W, K, layout), layout


def det_input_shape(h, w, limit=960):
    """ResizeImgType0 with limit_type 'max' (/root/reference/src/preprocess_op.cpp:74-88): the detector's input size of an
    h x w image.  C's round() (halves away from zero), float32 arithmetic as the reference."""
    ratio = np.float32(1.0) if max(h, w) <= limit else (np.float32(limit) / np.float32(h) if h > w else np.float32(limit) / np.float32(w))
    r32 = lambda v: int(np.floor(np.float32(v) / np.float32(32) + np.float32(0.5)))
    rh = max(r32(int(np.float32(h) * np.float32(ratio))) * 32, 32)
    rw = max(r32(int(np.float32(w) * np.float32(ratio))) * 32, 32)
    return rh, rw


def cfg3_item(i, limit=960):
    """cfg3 sample i with its probability map at the detector's input size (top-level: usable from worker processes)"""
    img = cfg3_sample(i)[0]
    rh, rw = det_input_shape(img.shape[0], img.shape[1], limit)
    return img, cfg3_prob_at(i, rh, rw)
