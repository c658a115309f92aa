"""Round-4 GPU parity tests (through the C-ABI): chunked ragged detector batches on a single detector lane, hostile
probability maps and the max_candidates cut, the OpenCV-version switch of the box score, capacity paths."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHUNK_CHILD = r"""
import sys, pickle
root, out = sys.argv[1], sys.argv[2]
sys.path[:0] = [root, root + "/oracle", root + "/tools"]
import numpy as np
from __graft_entry__ import load_package
from synth_data import cfg3_sample
pkg = load_package()
imgs = [cfg3_sample(i)[0] for i in range(5)]
kw = dict(rec_batch_num=16, rec_img_h=48, rec_img_w=320, enable_cls=True, limit_side_len=960)
res = {}
for phases in (1, 2):
    p = pkg.Pipe(phases=phases, **kw)
    got = p.run(imgs + [imgs[1]])
    res[phases] = [[(np.asarray(w["box"]).tolist(), np.asarray(w["ids"]).tolist(), float(w["confidence"])) for w in g] for g in got]
    p.close()
pickle.dump(res, open(out, "wb"))
print("CHILD OK")
"""


def _words(root_env, tmp_path, tag):
    out = str(tmp_path / (tag + ".pkl"))
    r = subprocess.run([sys.executable, "-c", _CHUNK_CHILD, ROOT, out], env=dict(os.environ, **root_env), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "CHILD OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    return pickle.load(open(out, "rb"))


@pytest.mark.parametrize("lanes", ["1", "2"])
def test_chunked_ragged_detector_with_one_lane_per_chain(built, tmp_path, lanes):
    """OCR_DET_LANES = 1 (or 2 with the default two chains) leaves a chain ONE detector lane;
    with more than one chunk per mixed-size batch (OCR_DET_CHUNK_MP = 1: every size group its own chunk) chunk k+1's
    network used to start on the instance that still held chunk k's maps.  Now a second instance exists whenever a batch
    has more than one chunk: words equal those of the default lanes / one 64 MP chunk, for one chain and for two."""
    want = _words({}, tmp_path, "default")
    got = _words({"OCR_DET_LANES": lanes, "OCR_DET_CHUNK_MP": "1"}, tmp_path, "lanes" + lanes)
    assert sum(len(g) for g in want[1]) > 0
    assert want[1] == want[2]
    assert got[1] == want[1] and got[2] == want[1]


def _same(bo, bg):
    return len(bo) == len(bg) and all(np.array_equal(a, b) for a, b in zip(bo, bg))


def _near_threshold_map(seed, H=480, W=640, n=150):
    """Rotated rectangles whose interior probabilities scatter around 0.5: with box_thresh = 0.5 the box score decides,
    and the pixels the two fillPoly rules disagree on (the span ends of the box mask, which lie OUTSIDE the blob for a
    rotated box) move many scores across the threshold."""
    rs = np.random.RandomState(seed)
    f = np.full((H, W), 0.02, np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    for _ in range(n):
        cx, cy = rs.uniform(20, W - 20), rs.uniform(20, H - 20)
        a, b, t = rs.uniform(6, 40), rs.uniform(3, 9), rs.uniform(-0.5, 0.5)
        u = (xx - cx) * np.cos(t) + (yy - cy) * np.sin(t)
        v = -(xx - cx) * np.sin(t) + (yy - cy) * np.cos(t)
        m = (np.abs(u) <= a) & (np.abs(v) <= b)
        f[m] = (rs.uniform(0.50, 0.62) + rs.uniform(-0.15, 0.15, int(m.sum()))).astype(np.float32)
    return f


@pytest.mark.parametrize("mode", ["fast", "slow"])
def test_box_score_follows_the_selected_opencv_fill_rule(pkg, built, mode):
    """ocr_det_cfg.cv_compat: cv::fillPoly's scan fill changed in OpenCV 4.5.2 (edges moved by half
    a pixel, spans [floor, floor] instead of [ceil, floor]); the mask feeds the `score < box_thresh` decision
    (postprocess_op.cpp:298), so box COUNTS depend on it.  Both rules, BoxScoreFast and PolygonScoreAcc: boxes == the
    oracle run with the same rule; and the rules really differ on these maps (the switch is not a no-op)."""
    import oracle as O
    slow = mode == "slow"
    counts = {}
    for compat in (45, 410):
        det = pkg.Det(limit_side_len=960, thresh=0.3, box_thresh=0.5, unclip_ratio=1.6, score_mode=mode, cv_compat=compat)
        tot = 0
        for seed in range(4):
            f = _near_threshold_map(seed)
            bo = O.det_post(f, 0.3, 0.5, 1.6, f.shape[0], f.shape[1], slow=slow, cv_compat=compat)
            bg = det.post(f, f.shape[0], f.shape[1])
            assert _same(bo, bg), (compat, seed, len(bo), len(bg))
            tot += len(bo)
        # a box poking out of the map: vertices outside the mask take the clipped-edge form of the 4.5.2+ rule
        g = np.full((96, 128), 0.02, np.float32)
        g[0:9, 0:70] = 0.55
        g[40:52, 100:128] = 0.6
        g[88:96, 30:90] = 0.52
        assert _same(O.det_post(g, 0.3, 0.5, 1.6, 96, 128, slow=slow, cv_compat=compat), det.post(g, 96, 128))
        counts[compat] = tot
        det.close()
    assert counts[45] > 0 and counts[410] > 0
    if not slow:
        assert counts[45] != counts[410], counts


def test_cv_compat_default_and_environment(pkg, built, monkeypatch):
    """cv_compat = 0 resolves to OCR_CV_COMPAT from the environment, else OCR_CV_410; anything else is refused."""
    import oracle as O
    f = _near_threshold_map(1)
    want = {c: O.det_post(f, 0.3, 0.5, 1.6, f.shape[0], f.shape[1], cv_compat=c) for c in (45, 410)}
    assert len(want[45]) != len(want[410])
    kw = dict(limit_side_len=960, thresh=0.3, box_thresh=0.5, unclip_ratio=1.6)
    d = pkg.Det(**kw)
    assert _same(want[410], d.post(f, *f.shape))
    d.close()
    monkeypatch.setenv("OCR_CV_COMPAT", "45")
    d = pkg.Det(**kw)
    assert _same(want[45], d.post(f, *f.shape))
    d.close()
    d = pkg.Det(cv_compat=410, **kw)          # an explicit value wins over the environment
    assert _same(want[410], d.post(f, *f.shape))
    d.close()
    monkeypatch.delenv("OCR_CV_COMPAT")
    with pytest.raises(Exception, match="cv_compat"):
        pkg.Det(cv_compat=300, **kw)


def test_candidate_cut_keeps_the_last_thousand_borders_on_the_device(pkg, built):
    """max_candidates = 1000 (postprocess_op.cpp:259-272): of 1100 blobs the reference keeps the FIRST 1000 contours
    findContours returns, i.e. the last 1000 borders its raster scan meets; same boxes, same order, no error."""
    import oracle as O
    H = W = 960
    f = np.full((H, W), 0.02, np.float32)
    k = 0
    for gy in range(34):
        for gx in range(33):
            if k == 1100:
                break
            y, x = 6 + gy * 28, 6 + gx * 29
            f[y:y + 14 + (k % 5), x:x + 18 + (k % 7)] = 0.9
            k += 1
    det = pkg.Det(limit_side_len=960, thresh=0.3, box_thresh=0.5, unclip_ratio=1.5)
    bo, bg = O.det_post(f, 0.3, 0.5, 1.5, H, W), det.post(f, H, W)
    assert len(bo) == 1000 and _same(bo, bg)
    # the survivors are the bottom-most blobs: nothing from the first rows of the grid
    assert min(int(b[:, 1].min()) for b in bg) > 6 + 2 * 28
    det.close()


@pytest.mark.parametrize("kind", ["salt0.5", "salt0.05", "checker", "stripes"])
@pytest.mark.timeout(600)
def test_hostile_maps_are_answered_like_the_reference(pkg, built, kind):
    """Noise maps with 10^4 - 10^5 borders: the reference simply keeps 1000 contours (postprocess_op.cpp:268-277) and
    answers; so does the device path - same boxes, no OCR_ERR_CAPACITY."""
    import oracle as O
    H = W = 960
    rs = np.random.RandomState(17)
    if kind.startswith("salt"):
        p = float(kind[4:])
        f = np.where(rs.rand(H, W) < p, 0.9, 0.02).astype(np.float32)
    elif kind == "checker":
        yy, xx = np.mgrid[0:H, 0:W]
        f = np.where((yy + xx) % 2 == 0, 0.9, 0.02).astype(np.float32)
    else:
        f = np.full((H, W), 0.02, np.float32)
        f[::2, :] = 0.9          # 480 one-pixel lines: every border a 2-vertex run, all dropped by the size gate
        f[:, 480] = 0.9          # ... joined into one comb
    for thr, mode in ((0.5, "fast"), (0.1, "fast"), (0.1, "slow")):
        det = pkg.Det(limit_side_len=960, thresh=0.3, box_thresh=thr, unclip_ratio=1.5, score_mode=mode)
        bo = O.det_post(f, 0.3, thr, 1.5, H, W, slow=mode == "slow")
        bg = det.post(f, H, W)
        assert _same(bo, bg), (kind, thr, mode, len(bo), len(bg))
        det.close()


@pytest.mark.parametrize("mode", ["fast", "slow"])
@pytest.mark.timeout(900)
def test_border_larger_than_the_default_key_pool_grows_the_pool(pkg, built, mode):
    """One component whose CHAIN_APPROX_SIMPLE border has more vertices than the default key pool holds (one-pixel
    zigzag lines every third row, joined by a spine: the chain turns at every pixel, 613 760 vertices against a pool of
    460 800 keys at 960 x 960): the first pass reports what it would have needed, the pools grow, the pass runs again -
    round 3 returned OCR_ERR_CAPACITY here (csrc/stages.hip), the reference answers (findContours allocates)."""
    import oracle as O
    H = W = 960
    f = np.full((H, W), 0.02, np.float32)
    xs = np.arange(W)
    for y in range(0, H - 1, 3):
        f[y + (xs % 2), xs] = 0.9
    f[:, 0] = 0.9
    det = pkg.Det(limit_side_len=960, thresh=0.3, box_thresh=0.1, unclip_ratio=1.5, score_mode=mode)
    bo = O.det_post(f, 0.3, 0.1, 1.5, H, W, slow=mode == "slow")
    bg = det.post(f, H, W)
    assert len(bo) >= 1 and _same(bo, bg)
    # and an ordinary map afterwards on the same handle: the default sizing is back, results unchanged
    g = _near_threshold_map(2)
    assert _same(O.det_post(g, 0.3, 0.1, 1.5, g.shape[0], g.shape[1], slow=mode == "slow"), det.post(g, *g.shape))
    det.close()


# ------------------------------------------------------------------------------------------------ OCR_ERR_CAPACITY
def test_every_buffer_too_small_path_reports_capacity_and_leaves_the_handle_usable(pkg, built, card):
    """OCR_ERR_CAPACITY (-4) = a CALLER buffer is too small (include/ocr_hip.h); every such return site is driven once:
    the call fails with the code and a message, nothing is written past the buffer, and the same handle answers the
    same request with a large enough buffer afterwards."""
    import ctypes as C
    import importlib
    from_binding = importlib.import_module(pkg.__name__ + ".binding")
    L = from_binding.lib()
    CAP = -4
    err = lambda: L.ocr_last_error().decode()
    # --- detector: more boxes than `cap` (stages.hip, "more boxes than the caller's capacity"), tap buffers (capi_stages.hip det_tap)
    det = pkg.Det(limit_side_len=960)
    full = det.run(card)
    assert len(full) > 2
    img = np.ascontiguousarray(card)
    boxes = np.full((2 + 1, 8), -7, np.int32)           # one guard row behind the 2-box buffer
    n = C.c_int()
    arr = from_binding._imgs([img])
    rc = L.ocr_det_run(det.h, arr, boxes.ctypes.data, 2, C.byref(n), None)
    assert rc == CAP and "capacity" in err() and n.value == len(full) and (boxes[2] == -7).all()
    assert np.array_equal(np.asarray(det.run(card)), np.asarray(full))
    _, h, w = det.last_shape()
    small = np.empty(16, np.float32)
    assert L.ocr_det_prob_map(det.h, 0, small.ctypes.data, small.size) == CAP and "too small" in err()
    assert L.ocr_det_bitmap(det.h, 0, small.ctypes.data, 16) == CAP
    assert L.ocr_det_resized(det.h, 0, small.ctypes.data, 16) == CAP
    assert det.prob_map().shape == (h, w)
    det.close()
    # --- recognizer: a text longer than max_len (stages.hip), step taps (capi_stages.hip ocr_rec_steps)
    rec = pkg.Rec(rec_img_h=48, rec_img_w=320)
    crop = np.ascontiguousarray(card[20:60, 10:300])
    ids, _ = rec.run([crop])
    assert len(ids[0]) > 1
    out = np.full(8, -7, np.int32)
    lens, sc = np.zeros(1, np.int32), np.zeros(1, np.float32)
    rc = L.ocr_rec_run(rec.h, from_binding._imgs([crop]), 1, out.ctypes.data, 1, lens.ctypes.data, sc.ctypes.data, None)
    assert rc == CAP and "max_len" in err() and (out[1:] == -7).all()
    again, _ = rec.run([crop])
    assert np.array_equal(again[0], ids[0])
    T = C.c_int()
    assert L.ocr_rec_steps(rec.h, 0, None, None, 1, C.byref(T)) == CAP and T.value > 1
    rec.close()
    # --- classifier tap (capi_stages.hip ocr_cls_probs)
    cls = pkg.Cls()
    cls.run([crop, crop])
    two = np.empty(2, np.float32)
    assert L.ocr_cls_probs(cls.h, two.ctypes.data, 2) == CAP
    assert cls.probs(2).shape == (2, 2)
    cls.close()
    # --- raw network taps (capi_net.hip ocr_net_fetch / ocr_net_timing_report)
    net = pkg.Net("cls")
    net.timing(True)
    net.forward(np.zeros((1, 48, 192, 3), np.float32))
    dims = (C.c_int * 4)()
    one = np.empty(1, np.float32)
    assert L.ocr_net_fetch(net.h, -1, one.ctypes.data, 1, dims) == CAP and dims[0] * dims[1] * dims[2] * dims[3] == 2
    buf = C.create_string_buffer(8)
    assert L.ocr_net_timing_report(net.h, buf, 8) == CAP and "report buffer" in err()
    assert len(net.timing_report()) > 10
    net.close()
    # --- pipeline: result buffers (pipe.hip emit), timing report (pipe.hip), rotate-crop buffer (pipe.hip), JPEG output (capi_jpeg.hip)
    pipe = pkg.Pipe()
    want = pipe.run([card])[0]
    assert len(want) > 2
    words = (from_binding.ocr_word * 2)()
    idsb = np.zeros(4096, np.int32)
    off, nw = np.zeros(1, np.int32), np.zeros(1, np.int32)
    rc = L.ocr_pipe_run(pipe.h, from_binding._imgs([img]), 1, words, 2, off.ctypes.data, nw.ctypes.data, idsb.ctypes.data, idsb.size, None)
    assert rc == CAP and "result buffers" in err()
    words = (from_binding.ocr_word * 64)()
    rc = L.ocr_pipe_run(pipe.h, from_binding._imgs([img]), 1, words, 64, off.ctypes.data, nw.ctypes.data, idsb.ctypes.data, 1, None)
    assert rc == CAP                                      # ids buffer too small
    got = pipe.run([card])[0]
    assert len(got) == len(want) and all(np.array_equal(a["ids"], b["ids"]) for a, b in zip(got, want))
    pipe.timing(True)
    pipe.run([card])
    assert L.ocr_pipe_timing_report(pipe.h, buf, 8) == CAP
    assert len(pipe.timing_report()) > 10
    pipe.close()
    box = np.array([[20, 20], [200, 24], [199, 60], [19, 56]], np.int32).reshape(-1)
    o_off = np.zeros(2, np.uint64)
    o_r, o_c = np.zeros(1, np.int32), np.zeros(1, np.int32)
    tiny = np.zeros(16, np.uint8)
    L.ocr_rotate_crop.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = L.ocr_rotate_crop(img.ctypes.data, img.shape[0], img.shape[1], img.strides[0], box.ctypes.data, 1, tiny.ctypes.data, tiny.size,
                           o_off.ctypes.data, o_r.ctypes.data, o_c.ctypes.data)
    assert rc == CAP and "crop buffer" in err() and o_r[0] > 0 and o_off[1] == 3 * o_r[0] * o_c[0] and not tiny.any()
    # one 8 x 8 grey block (DC only) through the device JPEG decoder: 192 output bytes
    class Comp(C.Structure):
        _fields_ = [("coef", C.c_void_p), ("quant", C.c_uint16 * 64), ("bw", C.c_int), ("bh", C.c_int), ("dw", C.c_int), ("dh", C.c_int)]

    class Jpeg(C.Structure):
        _fields_ = [("rows", C.c_int), ("cols", C.c_int), ("ncomp", C.c_int), ("hmax", C.c_int), ("vmax", C.c_int), ("comp", Comp * 3)]
    coef = np.zeros(64, np.int16)
    coef[0] = 40
    j = Jpeg(8, 8, 1, 1, 1)
    j.comp[0].coef = coef.ctypes.data
    for k in range(64):
        j.comp[0].quant[k] = 2
    j.comp[0].bw = j.comp[0].bh = 1
    j.comp[0].dw = j.comp[0].dh = 8
    L.ocr_jpeg_decode.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    out = np.full(192 + 8, 7, np.uint8)
    assert L.ocr_jpeg_decode(C.byref(j), 0, out.ctypes.data, 100) == CAP and (out == 7).all()
    assert L.ocr_jpeg_decode(C.byref(j), 0, out.ctypes.data, 192) == 0, err()
    assert (out[:192] == out[0]).all() and out[0] == 128 + 10 and (out[192:] == 7).all()   # 40 * 2 / 8 above mid-grey


# ------------------------------------------------------------------------------------------------ precision = "fp16"
def test_precision_parameter_fp16_is_accepted_and_int8_refused(pkg, built):
    """The reference's constructors take `precision` and hand it to TensorRT (ocr_det.cpp:50-56, ocr_cls.cpp:135-140,
    ocr_rec.cpp:167-172).  "fp32" and "fp16" build; anything else is an error with the parameter's name in it."""
    for mk in (lambda p: pkg.Det(precision=p), lambda p: pkg.Cls(precision=p), lambda p: pkg.Rec(precision=p), lambda p: pkg.Net("cls", precision=p)):
        h = mk("fp16")
        h.close()
        with pytest.raises(Exception, match="precision"):
            mk("int8")


def test_fp16_networks_stay_within_tolerance_of_the_f32_contract(pkg, built):
    """precision = "fp16": activation tensors stored as f16, matrix-core products in f16 with f32 accumulation; every VALU
    chain and reduction in f32 on values converted up.  Tolerances against the ORACLE (the f32
    contract), stated here:
      cls (the reference's real weights): softmax |d| <= 2e-3, labels identical;
      det (synthetic weights, logits up to ~10): probability map mean |d| <= 2e-3, 99 % of the pixels within 1e-2,
          thresholded bitmap (det_db_thresh 0.3) equal on >= 99.5 % of the pixels;
      rec (synthetic weights: 6625 nearly equal probabilities per step): |d| <= 2 % of the largest probability, arg max
          equal on >= 95 % of the steps."""
    from oracle import OracleNet
    rs = np.random.RandomState(2)
    x = rs.randn(6, 48, 192, 3).astype(np.float32)
    want, got = OracleNet("cls").run(x).reshape(6, 2), pkg.Net("cls", precision="fp16")
    y = got.forward(x).reshape(6, 2)
    got.close()
    assert np.abs(y - want).max() <= 2e-3 and np.array_equal(y.argmax(1), want.argmax(1))
    assert not np.array_equal(y, want)            # (it is another arithmetic: the fp32 mode is the bit-exact one)
    x = rs.randn(2, 160, 224, 3).astype(np.float32)
    want = OracleNet("det").run(x).reshape(-1)
    n = pkg.Net("det", precision="fp16")
    y = n.forward(x).reshape(-1)
    n.close()
    d = np.abs(y - want)
    assert d.mean() <= 2e-3 and np.quantile(d, 0.99) <= 1e-2, (d.mean(), np.quantile(d, 0.99), d.max())
    assert ((y > 0.3) == (want > 0.3)).mean() >= 0.995
    x = rs.randn(4, 48, 320, 3).astype(np.float32)
    want = OracleNet("rec").run(x).reshape(-1, 6625)
    n = pkg.Net("rec", precision="fp16")
    y = n.forward(x).reshape(-1, 6625)
    n.close()
    assert np.abs(y - want).max() <= 0.02 * want.max(), (np.abs(y - want).max(), want.max())
    assert (y.argmax(1) == want.argmax(1)).mean() >= 0.95


def test_fp16_stores_its_activation_tensors_as_f16(pkg, built):
    """precision = "fp16" keeps the C8I activation tensors in f16: every spatial tensor the classifier's
    launch list materialises holds only values a float16 represents exactly, in both the plain and the production launch list;
    the per-image vectors (pool results, SE gates), the logits and the probabilities stay f32; the fp32 mode's tensors are
    NOT f16 values (the storage follows the mode, the tensors are not merely small)."""
    rs = np.random.RandomState(4)
    x = rs.randn(3, 48, 192, 3).astype(np.float32)
    as_f16 = lambda t: np.array_equal(t, t.astype(np.float16).astype(np.float32))
    for keep in (1, 2):
        n16 = pkg.Net("cls", precision="fp16")
        n16.forward(x, keep_all=keep)
        spatial = vectors = 0
        for tid in range(1, n16.num_tensors()):
            if not n16.exists(tid):
                continue
            t = n16.fetch(tid)
            if t.shape[1] * t.shape[2] > 1 and t.shape[3] > 2:
                spatial += 1
                assert as_f16(t), (keep, tid, t.shape)
            elif t.shape[3] > 2:
                vectors += 1
        n16.close()
        assert spatial >= 20 and vectors >= 9, (spatial, vectors)
    n32 = pkg.Net("cls")
    n32.forward(x, keep_all=1)
    assert not as_f16(n32.fetch(3)) and not as_f16(n32.fetch(10))
    # a pooled vector of the fp16 mode is an f32 mean of f16 values: generally not an f16 value itself
    n16 = pkg.Net("cls", precision="fp16")
    n16.forward(x, keep_all=1)
    assert not as_f16(n16.fetch(4))
    n16.close()
    n32.close()


def test_fp16_stores_saturate_instead_of_overflowing(pkg, built):
    """An activation beyond the f16 range is stored as +-65504, not as infinity (conv_device.h, st4): a network fed values a
    hundred thousand times larger than a normalised image still ends in finite probabilities that sum to one - an inf in a
    tensor would reach the softmax as NaN through the next layer's 0 * inf or inf - inf - and an intermediate tensor of that
    run holds the saturation value itself."""
    rs = np.random.RandomState(6)
    x = (rs.randn(2, 48, 192, 3) * 1e5).astype(np.float32)
    n = pkg.Net("cls", precision="fp16")
    y = n.forward(x, keep_all=1).reshape(2, 2)
    assert np.isfinite(y).all() and np.allclose(y.sum(1), 1.0, atol=1e-3), y
    stem = n.fetch(1)
    assert np.isfinite(stem).all() and np.abs(stem).max() == 65504.0
    n.close()


_FP16_X8_CHILD = r"""
import sys, os
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
from __graft_entry__ import load_package
pkg = load_package()
rs = np.random.RandomState(12)
out = {}
for kind, shape in (("cls", (4, 48, 192)), ("det", (2, 160, 224)), ("rec", (3, 48, 320))):
    x = rs.randn(shape[0], shape[1], shape[2], 3).astype(np.float32)
    a, b = pkg.Net(kind), pkg.Net(kind, precision="fp16")
    ya, yb = a.forward(x).reshape(-1), b.forward(x).reshape(-1)
    a.close(); b.close()
    d = np.abs(ya - yb)
    print(kind, float(d.mean()), float(d.max()), float(np.abs(ya).max()))
    assert np.isfinite(yb).all()
    if kind == "cls": assert d.max() <= 2e-3
    if kind == "det": assert d.mean() <= 2e-3 and np.quantile(d, 0.99) <= 1e-2
    if kind == "rec": assert d.max() <= 0.02 * np.abs(ya).max()
print("X8 OK")
"""


def test_fp16_with_the_32x32x8_kernels_keeps_the_tolerances(built):
    """OCR_MFMA_X16=0 (read once per process: a child process) sends the fp16 mode's big 1x1 convs and 3x3 96-channel convs back
    to their v_mfma_f32_32x32x8_f16 forms - the instantiations every other shape of the mode runs anyway; the same tolerances
    against the fp32 mode hold."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", _FP16_X8_CHILD, root], env=dict(os.environ, OCR_MFMA_X16="0"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "X8 OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_fp16_pipeline_agrees_with_fp32_on_the_benchmark_batch(pkg, built):
    """The whole path in fp16 on 8 configs[1] images (256 lines, probability-map protocol): boxes are identical (they
    come from the protocol's maps), the CTC id sequences agree on >= 95 % of the lines, confidences within 1e-4; the
    ragged recognizer batch, the fused depthwise blocks and the gated 1x1 convs all run their f16 instantiations."""
    from synth_data import cfg2_sample
    imgs, probs = zip(*[cfg2_sample(i)[:2] for i in range(8)])
    kw = dict(enable_cls=True, limit_side_len=960, rec_batch_num=16, rec_img_h=48, rec_img_w=320)
    words = {}
    for prec in ("fp32", "fp16"):
        p = pkg.Pipe(precision=prec, **kw)
        d_i, d_p = pkg.DevArray(np.stack(imgs)), pkg.DevArray(np.stack(probs))
        words[prec] = p.run_device(d_i, 960, 960, 8, d_p, collect=True)
        p.close()
    tot = same = 0
    for wa, wb in zip(words["fp32"], words["fp16"]):
        assert len(wa) == len(wb) == 32
        for a, b in zip(wa, wb):
            assert np.array_equal(a["box"], b["box"]) and abs(a["confidence"] - b["confidence"]) <= 1e-4
            tot += 1
            same += bool(np.array_equal(a["ids"], b["ids"]))
    assert same / tot >= 0.95, same / tot


def test_fp16_on_mixed_sizes_and_the_worker_shape(pkg, built, card):
    """The fp16 instantiations on the other launch lists: a ragged batch of IMAGES (the detector on mixed sizes: every
    dense conv through the direct kernel, the fused depthwise blocks in their ragged form) - probability maps within the
    det tolerance of the fp32 maps, image by image - and the worker's default shape (limit 512, rec 28 x 192) on the
    reference's card image: same number of boxes within one, the pipeline runs to words."""
    from synth_data import cfg3_sample
    rs = np.random.RandomState(9)
    imgs = [rs.randn(h, w, 3).astype(np.float32) for h, w in ((96, 160), (64, 64), (160, 96), (128, 224))]
    a, b = pkg.Net("det"), pkg.Net("det", precision="fp16")
    ya, yb = a.forward_ragged_images(imgs).reshape(-1), b.forward_ragged_images(imgs).reshape(-1)
    a.close()
    b.close()
    d = np.abs(ya - yb)
    assert d.mean() <= 2e-3 and np.quantile(d, 0.99) <= 1e-2 and ((ya > 0.3) == (yb > 0.3)).mean() >= 0.995
    lines = [rs.randn(48, w, 3).astype(np.float32) for w in (320, 333, 136, 320)]
    a, b = pkg.Net("rec"), pkg.Net("rec", precision="fp16")
    ya, yb = a.forward_ragged(lines).reshape(-1, 6625), b.forward_ragged(lines).reshape(-1, 6625)
    a.close()
    b.close()
    assert np.abs(ya - yb).max() <= 0.02 * ya.max() and (ya.argmax(1) == yb.argmax(1)).mean() >= 0.95
    p32, p16 = pkg.Pipe(), pkg.Pipe(precision="fp16")
    w32, w16 = p32.run([card])[0], p16.run([card])[0]
    mixed = [cfg3_sample(i)[0] for i in range(3)]
    m16 = p16.run(mixed)
    p32.close()
    p16.close()
    assert abs(len(w32) - len(w16)) <= 1 and len(w16) > 0 and len(m16) == 3


def test_pool_growth_inside_a_ragged_detector_chunk_and_cv_compat_through_the_pipeline(pkg, built):
    """The mixed-size path (DetStage::post_mixed: one pass of the post kernels over a chunk of images of different sizes)
    shares post_launch with the uniform path: an image whose borders outgrow the default key pool sits in the SAME chunk as
    ordinary images - the pass is re-run with grown pools and every image's words still equal a run of the image alone.
    And ocr_pipe_cfg.det.cv_compat reaches the chains' detector instances: box counts follow the selected fill rule."""
    from pipeline import Pipeline, DetCfg
    from synth_data import cfg3_item
    zig = np.full((960, 960), 0.02, np.float32)     # the 613 760-vertex border of the test above, as image 1's probability map
    xs = np.arange(960)
    for y in range(0, 959, 3):
        zig[y + (xs % 2), xs] = 0.9
    zig[:, 0] = 0.9
    items = [cfg3_item(0), (np.random.RandomState(1).randint(0, 255, (960, 960, 3)).astype(np.uint8), zig), cfg3_item(1)]
    imgs, probs = [it[0] for it in items], [it[1] for it in items]
    assert len({im.shape for im in imgs}) == 3
    kw = dict(rec_batch_num=16, rec_img_h=48, rec_img_w=320, limit_side_len=960, box_thresh=0.1)
    p = pkg.Pipe(**kw)
    p.stage(0, imgs, probs)
    together = p.run_staged(0)
    alone = []
    for im, pr in zip(imgs, probs):
        p.stage(1, [im], [pr])
        alone.append(p.run_staged(1)[0])
    p.close()
    assert len(together[1]) >= 1 and len(together[0]) > 3
    for a, b in zip(together, alone):
        assert len(a) == len(b) and all(np.array_equal(x["box"], y["box"]) and np.array_equal(x["ids"], y["ids"]) for x, y in zip(a, b))
    kw.pop("box_thresh")
    # the oracle on the first image under both OpenCV rules, against pipelines created with each
    for compat in (45, 410):
        pg = pkg.Pipe(cv_compat=compat, **kw)
        po = Pipeline(det_cfg=DetCfg(limit_side_len=960, cv_compat=compat), rec_batch_num=16, rec_img_h=48, rec_img_w=320)
        g, w = pg.run([imgs[0]])[0], po.process(imgs[0])["words"]
        pg.close()
        assert len(g) == len(w) and all(np.array_equal(a["box"], b["box"]) and np.array_equal(a["ids"], b["ids"]) for a, b in zip(g, w))
