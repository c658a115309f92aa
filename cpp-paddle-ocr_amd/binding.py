"""ctypes binding of include/ocr_hip.h (plumbing for tests and bench.py).

Fails loudly when the HIP library is missing or no gfx950 device is visible: there is no
CPU fallback anywhere in the product path.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_PATH = os.environ.get("OCR_LIB_PATH") or os.path.join(HERE, "lib", "libocr_hip.so")   # (override: A/B runs against another build)
MODELS = os.path.join(ROOT, "models")

_lib = None


class OcrError(RuntimeError):
    pass


class ocr_img(C.Structure):
    _fields_ = [("data", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("row_stride", C.c_size_t)]


class ocr_det_cfg(C.Structure):
    _fields_ = [("model_dir", C.c_char_p), ("device_id", C.c_int), ("limit_type", C.c_char_p),
                ("limit_side_len", C.c_int), ("det_db_thresh", C.c_double), ("det_db_box_thresh", C.c_double),
                ("det_db_unclip_ratio", C.c_double), ("det_db_score_mode", C.c_char_p), ("use_dilation", C.c_int),
                ("precision", C.c_char_p), ("max_batch", C.c_int), ("cv_compat", C.c_int)]


CV_45, CV_410 = 45, 410  # ocr_det_cfg.cv_compat (0 = default = OCR_CV_COMPAT from the environment, else CV_410)


class ocr_cls_cfg(C.Structure):
    _fields_ = [("model_dir", C.c_char_p), ("device_id", C.c_int), ("cls_thresh", C.c_double),
                ("cls_batch_num", C.c_int), ("precision", C.c_char_p)]


class ocr_rec_cfg(C.Structure):
    _fields_ = [("model_dir", C.c_char_p), ("device_id", C.c_int), ("label_path", C.c_char_p),
                ("rec_batch_num", C.c_int), ("rec_img_h", C.c_int), ("rec_img_w", C.c_int),
                ("precision", C.c_char_p), ("sort_mode", C.c_int)]


# every symbol include/ocr_hip.h declares (tests check the .so exports all of them)
EXPORTS = [
    "ocr_last_error", "ocr_rt_init", "ocr_rt_device_count", "ocr_rt_set_wait_mode", "ocr_rt_get_wait_mode",
    "ocr_det_cfg_default", "ocr_det_create", "ocr_det_destroy", "ocr_det_run", "ocr_det_run_batch",
    "ocr_det_last_shape", "ocr_det_prob_map", "ocr_det_bitmap", "ocr_det_resized", "ocr_det_post",
    "ocr_cls_cfg_default", "ocr_cls_create", "ocr_cls_destroy", "ocr_cls_run", "ocr_cls_probs",
    "ocr_rec_cfg_default", "ocr_rec_create", "ocr_rec_destroy", "ocr_rec_run", "ocr_rec_label",
    "ocr_rec_num_classes", "ocr_rec_steps",
    "ocr_net_create", "ocr_net_create_precision", "ocr_net_destroy", "ocr_net_forward", "ocr_net_forward_ragged", "ocr_net_forward_ragged_images", "ocr_net_num_tensors", "ocr_net_tensor_exists", "ocr_net_fetch",
    "ocr_net_timing", "ocr_net_timing_report", "ocr_probe", "ocr_selftest_refuse_launch", "ocr_selftest_lds_memo",
    "ocr_selftest_unclip", "ocr_selftest_unclip_box",
    "ocr_srv_net_create", "ocr_srv_net_destroy", "ocr_srv_net_forward", "ocr_srv_net_rerun", "ocr_srv_net_num_tensors", "ocr_srv_net_fetch",
    "ocr_srv_net_timing", "ocr_srv_net_timing_report",
]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OcrError("libocr_hip.so is not built (%s): run __graft_entry__.build()" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.ocr_last_error.restype = C.c_char_p
        L.ocr_net_create.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
        L.ocr_net_create_precision.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]
        L.ocr_net_destroy.argtypes = [C.c_void_p]
        L.ocr_net_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        if hasattr(L, "ocr_net_forward_ragged_images"):
            L.ocr_net_forward_ragged_images.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        if hasattr(L, "ocr_net_forward_ragged"):
            L.ocr_net_forward_ragged.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.ocr_net_num_tensors.argtypes = [C.c_void_p]
        L.ocr_net_tensor_exists.argtypes = [C.c_void_p, C.c_int]
        L.ocr_net_fetch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]
        L.ocr_net_timing.argtypes = [C.c_void_p, C.c_int]
        L.ocr_net_timing_report.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.ocr_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        if hasattr(L, "ocr_selftest_refuse_launch"):
            L.ocr_selftest_refuse_launch.argtypes = [C.c_char_p]
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise OcrError("libocr_hip error %d: %s" % (rc, lib().ocr_last_error().decode(errors="replace")))


class SrvNet:
    """raw taps of a server network (BASELINE configs[4]; hand-written plans, seeded weights): kind "det" | "rec" """

    def __init__(self, kind, precision="fp16", model_dir=None, device=0):
        L = lib()
        L.ocr_srv_net_create.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]
        L.ocr_srv_net_destroy.argtypes = [C.c_void_p]
        L.ocr_srv_net_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ocr_srv_net_rerun.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ocr_srv_net_num_tensors.argtypes = [C.c_void_p]
        L.ocr_srv_net_fetch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]
        L.ocr_srv_net_timing.argtypes = [C.c_void_p, C.c_int]
        L.ocr_srv_net_timing_report.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        if model_dir is None:
            root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            model_dir = os.path.join(root, "models_server", kind)
        self.h = C.c_void_p()
        check(L.ocr_srv_net_create(kind.encode(), model_dir.encode(), device, precision.encode(), C.byref(self.h)))
        self.shape = None

    def forward(self, x, keep_all=False):
        x = np.ascontiguousarray(x, dtype=np.float32)
        n, h, w, c = x.shape
        assert c == 3
        self.shape = (n, h, w)
        check(lib().ocr_srv_net_forward(self.h, x.ctypes.data, n, h, w, 1 if keep_all else 0))
        return self.fetch(-1)

    def rerun(self, iters=1):
        n, h, w = self.shape
        check(lib().ocr_srv_net_rerun(self.h, n, h, w, iters))

    def num_tensors(self):
        return lib().ocr_srv_net_num_tensors(self.h)

    def fetch(self, tid, cap=None):
        dims = (C.c_int * 4)()
        cap = cap or (1 << 22)
        while True:
            out = np.empty(cap, np.float32)
            rc = lib().ocr_srv_net_fetch(self.h, tid, out.ctypes.data, cap, dims)
            if rc == -4:
                cap *= 8
                continue
            check(rc)
            break
        n = dims[0] * dims[1] * dims[2] * dims[3]
        return out[:n].reshape(dims[0], dims[1], dims[2], dims[3]).copy()

    def timing(self, on=True):
        check(lib().ocr_srv_net_timing(self.h, 1 if on else 0))

    def timing_report(self):
        buf = C.create_string_buffer(1 << 18)
        check(lib().ocr_srv_net_timing_report(self.h, buf, len(buf)))
        rep = {}
        for line in buf.value.decode().splitlines():
            name, ms, cnt, fl, by = line.rsplit(" ", 4)
            rep[name] = dict(ms=float(ms), count=int(cnt), flops=float(fl), bytes=float(by))
        return rep

    def close(self):
        if self.h:
            lib().ocr_srv_net_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def selftest_unclip(quads, deltas, cap=64):
    """device ClipperOffset (jtRound, closed polygon) of n int quads -> list of [k][2] int64 arrays (None: scratch overflow)"""
    quads = np.ascontiguousarray(quads, dtype=np.int32).reshape(-1, 8)
    deltas = np.ascontiguousarray(deltas, dtype=np.float64)
    n = quads.shape[0]
    out = np.zeros((n, cap, 2), np.int64)
    counts = np.zeros(n, np.int32)
    trig = np.zeros((n, 3), np.float64)
    L = lib()
    L.ocr_selftest_unclip.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    check(L.ocr_selftest_unclip(quads.ctypes.data, deltas.ctypes.data, n, out.ctypes.data, cap, counts.ctypes.data, trig.ctypes.data))
    return out, counts, trig


def selftest_unclip_box(boxes, unclip_ratio):
    """device UnClip -> minAreaRect -> GetMiniBoxes of n float boxes -> (out14 [n,14], status [n,2])"""
    boxes = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 8)
    n = boxes.shape[0]
    out = np.zeros((n, 14), np.float32)
    st = np.zeros((n, 2), np.int32)
    L = lib()
    L.ocr_selftest_unclip_box.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
    check(L.ocr_selftest_unclip_box(boxes.ctypes.data, float(unclip_ratio), n, out.ctypes.data, st.ctypes.data))
    return out, st


def probe(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    out = np.empty((8, a.size), dtype=np.float32)
    check(lib().ocr_probe(a.ctypes.data, b.ctypes.data, out.ctypes.data, a.size))
    return out


class Net:
    """Raw network tap (ocr_net_*): host f32 NHWC in, logical NHWC tensors out."""

    def __init__(self, kind, model_dir=None, weights=None, device=0, precision="fp32"):
        self.kind = kind
        model_dir = model_dir or os.path.join(MODELS, kind)
        self.h = C.c_void_p()
        check(lib().ocr_net_create_precision(kind.encode(), model_dir.encode(), weights.encode() if weights else None, device,
                                             precision.encode(), C.byref(self.h)))

    def forward(self, x_nhwc, keep_all=False):
        x = np.ascontiguousarray(x_nhwc, dtype=np.float32)
        n, h, w, c = x.shape
        assert c == 3
        check(lib().ocr_net_forward(self.h, x.ctypes.data, n, h, w, int(keep_all)))
        return self.fetch(-1)

    def forward_ragged(self, lines, keep_all=False):
        """lines: list of f32 [H, W_i, 3] arrays of one height -> the output's lines one after the other [1, 1, sum T_i, C]"""
        h = lines[0].shape[0]
        assert all(l.shape[0] == h and l.shape[2] == 3 for l in lines)
        x = np.concatenate([np.ascontiguousarray(l, dtype=np.float32).reshape(-1) for l in lines])
        widths = np.array([l.shape[1] for l in lines], dtype=np.int32)
        check(lib().ocr_net_forward_ragged(self.h, x.ctypes.data, len(lines), h, widths.ctypes.data, int(keep_all)))
        return self.fetch(-1)

    def forward_ragged_images(self, imgs, keep_all=False):
        """imgs: list of f32 [H_i, W_i, 3] arrays (sizes multiples of 32) -> the output's images one after the other"""
        x = np.concatenate([np.ascontiguousarray(l, dtype=np.float32).reshape(-1) for l in imgs])
        hs = np.array([l.shape[0] for l in imgs], dtype=np.int32)
        ws = np.array([l.shape[1] for l in imgs], dtype=np.int32)
        check(lib().ocr_net_forward_ragged_images(self.h, x.ctypes.data, len(imgs), hs.ctypes.data, ws.ctypes.data, int(keep_all)))
        return self.fetch(-1)

    def fetch(self, tid, cap=None):
        dims = (C.c_int * 4)()
        cap = cap or (1 << 28)
        # first call with a tiny buffer is not possible (no size query) -> allocate by trying sizes
        buf = np.empty(min(cap, 1 << 22), dtype=np.float32)
        rc = lib().ocr_net_fetch(self.h, tid, buf.ctypes.data, buf.size, dims)
        if rc == -4:
            n = dims[0] * dims[1] * dims[2] * dims[3]
            buf = np.empty(n, dtype=np.float32)
            rc = lib().ocr_net_fetch(self.h, tid, buf.ctypes.data, buf.size, dims)
        check(rc)
        n = dims[0] * dims[1] * dims[2] * dims[3]
        return buf[:n].reshape(dims[0], dims[1], dims[2], dims[3]).copy()

    def num_tensors(self):
        return lib().ocr_net_num_tensors(self.h)

    def exists(self, tid):
        """did the last forward write plan tensor `tid`? (a fused-away tensor never reaches device memory)"""
        return bool(lib().ocr_net_tensor_exists(self.h, int(tid)))

    def timing(self, on=True):
        check(lib().ocr_net_timing(self.h, int(on)))

    def timing_report(self):
        buf = C.create_string_buffer(1 << 16)
        check(lib().ocr_net_timing_report(self.h, buf, len(buf)))
        out = {}
        for line in buf.value.decode().splitlines():
            name, ms, cnt, fl, by = line.split()
            out[name] = dict(ms=float(ms), count=int(cnt), flops=float(fl), bytes=float(by))
        return out

    def close(self):
        if self.h:
            lib().ocr_net_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------
# stage handles
# ---------------------------------------------------------------------------------------------
def _img(a):
    a = np.asarray(a)
    if a.size == 0:
        return ocr_img(None, 0, 0, 0)  # the library reports "Empty image data provided"
    assert a.dtype == np.uint8 and a.ndim == 3 and a.shape[2] == 3 and a.strides[2] == 1 and a.strides[1] == 3
    return ocr_img(a.ctypes.data, a.shape[0], a.shape[1], a.strides[0])


def _imgs(arrs):
    arr = (ocr_img * len(arrs))()
    for i, a in enumerate(arrs):
        arr[i] = _img(a)
    return arr


def _stage_protos(L):
    if getattr(L, "_stage_protos_done", False):
        return
    vp, ip = C.c_void_p, C.POINTER(C.c_int)
    L.ocr_det_cfg_default.argtypes = [C.POINTER(ocr_det_cfg)]
    L.ocr_det_create.argtypes = [C.POINTER(ocr_det_cfg), C.POINTER(vp)]
    L.ocr_det_destroy.argtypes = [vp]
    L.ocr_det_run.argtypes = [vp, C.POINTER(ocr_img), vp, C.c_int, ip, C.POINTER(C.c_double)]
    L.ocr_det_run_batch.argtypes = [vp, C.POINTER(ocr_img), C.c_int, vp, C.c_int, vp, C.POINTER(C.c_double)]
    L.ocr_det_last_shape.argtypes = [vp, ip, ip, ip]
    L.ocr_det_prob_map.argtypes = [vp, C.c_int, vp, C.c_size_t]
    L.ocr_det_bitmap.argtypes = [vp, C.c_int, vp, C.c_size_t]
    L.ocr_det_resized.argtypes = [vp, C.c_int, vp, C.c_size_t]
    L.ocr_det_post.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, ip]
    L.ocr_cls_cfg_default.argtypes = [C.POINTER(ocr_cls_cfg)]
    L.ocr_cls_create.argtypes = [C.POINTER(ocr_cls_cfg), C.POINTER(vp)]
    L.ocr_cls_destroy.argtypes = [vp]
    L.ocr_cls_run.argtypes = [vp, C.POINTER(ocr_img), C.c_int, vp, vp, C.POINTER(C.c_double)]
    L.ocr_cls_probs.argtypes = [vp, vp, C.c_size_t]
    L.ocr_rec_cfg_default.argtypes = [C.POINTER(ocr_rec_cfg)]
    L.ocr_rec_create.argtypes = [C.POINTER(ocr_rec_cfg), C.POINTER(vp)]
    L.ocr_rec_destroy.argtypes = [vp]
    L.ocr_rec_run.argtypes = [vp, C.POINTER(ocr_img), C.c_int, vp, C.c_int, vp, vp, C.POINTER(C.c_double)]
    L.ocr_rec_label.argtypes = [vp, C.c_int]
    L.ocr_rec_label.restype = C.c_char_p
    L.ocr_rec_num_classes.argtypes = [vp]
    L.ocr_rec_steps.argtypes = [vp, C.c_int, vp, vp, C.c_int, ip]
    L._stage_protos_done = True


class Det:
    """DBDetector over the C-ABI (ocr_det_*).  Defaults = the literals OCRWorker passes."""

    def __init__(self, model_dir=None, device=0, limit_type="max", limit_side_len=512, thresh=0.2, box_thresh=0.4,
                 unclip_ratio=1.8, score_mode="fast", use_dilation=False, precision="fp32", max_batch=1, cv_compat=0):
        L = lib()
        _stage_protos(L)
        cfg = ocr_det_cfg()
        L.ocr_det_cfg_default(C.byref(cfg))
        self._keep = [(model_dir or os.path.join(MODELS, "det")).encode(), limit_type.encode(), score_mode.encode(),
                      precision.encode()]
        cfg.model_dir, cfg.limit_type, cfg.det_db_score_mode, cfg.precision = self._keep
        cfg.device_id, cfg.limit_side_len = device, limit_side_len
        cfg.det_db_thresh, cfg.det_db_box_thresh, cfg.det_db_unclip_ratio = thresh, box_thresh, unclip_ratio
        cfg.use_dilation, cfg.max_batch = int(use_dilation), max_batch
        cfg.cv_compat = int(cv_compat)
        self.h = C.c_void_p()
        check(L.ocr_det_create(C.byref(cfg), C.byref(self.h)))
        self.times = (C.c_double * 3)()

    def run(self, img, cap=2000):
        return self.run_batch([img], cap)[0]

    def run_batch(self, imgs, cap=2000):
        n = len(imgs)
        boxes = np.zeros((n, cap, 8), np.int32)
        cnt = np.zeros(n, np.int32)
        arr = _imgs(imgs)
        check(lib().ocr_det_run_batch(self.h, arr, n, boxes.ctypes.data, cap, cnt.ctypes.data, self.times))
        return [boxes[i, :cnt[i]].reshape(-1, 4, 2).copy() for i in range(n)]

    def last_shape(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        check(lib().ocr_det_last_shape(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def prob_map(self, index=0):
        _, h, w = self.last_shape()
        out = np.empty((h, w), np.float32)
        check(lib().ocr_det_prob_map(self.h, index, out.ctypes.data, out.size))
        return out

    def bitmap(self, index=0):
        _, h, w = self.last_shape()
        out = np.empty((h, w), np.uint8)
        check(lib().ocr_det_bitmap(self.h, index, out.ctypes.data, out.size))
        return out

    def resized(self, index=0):
        _, h, w = self.last_shape()
        out = np.empty((h, w, 3), np.uint8)
        check(lib().ocr_det_resized(self.h, index, out.ctypes.data, out.size))
        return out

    def post(self, prob, src_h, src_w, cap=2000):
        prob = np.ascontiguousarray(prob, dtype=np.float32)
        boxes = np.zeros((cap, 8), np.int32)
        n = C.c_int()
        check(lib().ocr_det_post(self.h, prob.ctypes.data, prob.shape[0], prob.shape[1], src_h, src_w,
                                 boxes.ctypes.data, cap, C.byref(n)))
        return boxes[:n.value].reshape(-1, 4, 2).copy()

    def close(self):
        if self.h:
            lib().ocr_det_destroy(self.h)
            self.h = None


class Cls:
    def __init__(self, model_dir=None, device=0, cls_thresh=0.98, cls_batch_num=8, precision="fp32"):
        L = lib()
        _stage_protos(L)
        cfg = ocr_cls_cfg()
        L.ocr_cls_cfg_default(C.byref(cfg))
        self._keep = [(model_dir or os.path.join(MODELS, "cls")).encode(), precision.encode()]
        cfg.model_dir, cfg.precision = self._keep
        cfg.device_id, cfg.cls_thresh, cfg.cls_batch_num = device, cls_thresh, cls_batch_num
        self.h = C.c_void_p()
        check(L.ocr_cls_create(C.byref(cfg), C.byref(self.h)))
        self.times = (C.c_double * 3)()

    def run(self, crops):
        n = len(crops)
        labels = np.zeros(n, np.int32)
        scores = np.zeros(n, np.float32)
        if n:
            check(lib().ocr_cls_run(self.h, _imgs(crops), n, labels.ctypes.data, scores.ctypes.data, self.times))
        return labels, scores

    def probs(self, n):
        out = np.empty((n, 2), np.float32)
        check(lib().ocr_cls_probs(self.h, out.ctypes.data, out.size))
        return out

    def close(self):
        if self.h:
            lib().ocr_cls_destroy(self.h)
            self.h = None


class Rec:
    def __init__(self, model_dir=None, device=0, label_path=None, rec_batch_num=16, rec_img_h=28, rec_img_w=192,
                 precision="fp32", sort_mode=0):
        L = lib()
        _stage_protos(L)
        cfg = ocr_rec_cfg()
        L.ocr_rec_cfg_default(C.byref(cfg))
        md = model_dir or os.path.join(MODELS, "rec")
        self._keep = [md.encode(), (label_path or os.path.join(md, "ppocr_keys_v1.txt")).encode(), precision.encode()]
        cfg.model_dir, cfg.label_path, cfg.precision = self._keep
        cfg.device_id, cfg.rec_batch_num, cfg.rec_img_h, cfg.rec_img_w = device, rec_batch_num, rec_img_h, rec_img_w
        cfg.sort_mode = int(sort_mode)
        self.h = C.c_void_p()
        check(L.ocr_rec_create(C.byref(cfg), C.byref(self.h)))
        self.times = (C.c_double * 3)()

    def run(self, crops, max_len=512):
        n = len(crops)
        ids = np.zeros((n, max_len), np.int32)
        lens = np.zeros(n, np.int32)
        scores = np.zeros(n, np.float32)
        if n:
            check(lib().ocr_rec_run(self.h, _imgs(crops), n, ids.ctypes.data, max_len, lens.ctypes.data,
                                    scores.ctypes.data, self.times))
        return [ids[i, :lens[i]].copy() for i in range(n)], scores

    def steps(self, index, cap=4096):
        amax = np.zeros(cap, np.int32)
        pmax = np.zeros(cap, np.float32)
        T = C.c_int()
        check(lib().ocr_rec_steps(self.h, index, amax.ctypes.data, pmax.ctypes.data, cap, C.byref(T)))
        return amax[:T.value].copy(), pmax[:T.value].copy()

    def label(self, i):
        s = lib().ocr_rec_label(self.h, int(i))
        return s.decode("utf-8") if s is not None else None

    def text(self, ids):
        return "".join(self.label(i) for i in ids)

    def num_classes(self):
        return lib().ocr_rec_num_classes(self.h)

    def close(self):
        if self.h:
            lib().ocr_rec_destroy(self.h)
            self.h = None


# ---------------------------------------------------------------------------------------------
# fused pipeline + device helpers
# ---------------------------------------------------------------------------------------------
class ocr_pipe_cfg(C.Structure):
    _fields_ = [("det", ocr_det_cfg), ("cls", ocr_cls_cfg), ("rec", ocr_rec_cfg), ("enable_cls", C.c_int),
                ("crop_mode", C.c_int), ("phases", C.c_int)]


class ocr_word(C.Structure):
    _fields_ = [("box", C.c_int32 * 8), ("ids_off", C.c_int32), ("ids_len", C.c_int32), ("confidence", C.c_float)]


EXPORTS += ["ocr_pipe_cfg_default", "ocr_pipe_create", "ocr_pipe_destroy", "ocr_pipe_run", "ocr_pipe_run_device",
            "ocr_pipe_stage", "ocr_pipe_slot_probs", "ocr_pipe_run_staged", "ocr_pipe_run_device_on", "ocr_pipe_run_staged_on", "ocr_pipe_stage_jpeg", "ocr_jpeg_decode",
            "ocr_pipe_label", "ocr_pipe_det_shape", "ocr_pipe_stats", "ocr_pipe_timing", "ocr_pipe_timing_filter", "ocr_pipe_timing_report", "ocr_dev_alloc",
            "ocr_dev_free", "ocr_dev_upload", "ocr_dev_download", "ocr_dev_sync", "ocr_rotate_crop", "ocr_rotate_crop_shape", "ocr_rotate180_rois"]


def _pipe_protos(L):
    if getattr(L, "_pipe_protos_done", False):
        return
    _stage_protos(L)
    vp, ip = C.c_void_p, C.POINTER(C.c_int)
    L.ocr_pipe_cfg_default.argtypes = [C.POINTER(ocr_pipe_cfg)]
    L.ocr_pipe_create.argtypes = [C.POINTER(ocr_pipe_cfg), C.POINTER(vp)]
    L.ocr_pipe_destroy.argtypes = [vp]
    L.ocr_pipe_run.argtypes = [vp, C.POINTER(ocr_img), C.c_int, vp, C.c_int, vp, vp, vp, C.c_int, C.POINTER(C.c_double)]
    L.ocr_pipe_run_device.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp, vp, C.c_int,
                                      C.POINTER(C.c_double)]
    L.ocr_pipe_stage.argtypes = [vp, C.c_int, C.POINTER(ocr_img), C.c_int]
    L.ocr_pipe_slot_probs.argtypes = [vp, C.c_int, vp, C.c_int]
    L.ocr_pipe_run_staged.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp, vp, C.c_int, C.POINTER(C.c_double)]
    L.ocr_pipe_run_staged_on.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, vp, vp, vp, C.c_int, C.POINTER(C.c_double)]
    L.ocr_pipe_run_device_on.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp, vp, C.c_int,
                                         C.POINTER(C.c_double)]
    L.ocr_pipe_label.argtypes = [vp, C.c_int]
    L.ocr_pipe_label.restype = C.c_char_p
    L.ocr_pipe_det_shape.argtypes = [vp, C.c_int, C.c_int, ip, ip]
    L.ocr_pipe_timing.argtypes = [vp, C.c_int]
    L.ocr_pipe_stats.argtypes = [vp, C.POINTER(C.c_longlong)]
    L.ocr_pipe_timing_report.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.ocr_pipe_timing_filter.argtypes = [vp, C.c_char_p]
    L.ocr_dev_alloc.argtypes = [C.POINTER(vp), C.c_size_t]
    L.ocr_dev_free.argtypes = [vp]
    L.ocr_dev_upload.argtypes = [vp, vp, C.c_size_t]
    L.ocr_dev_download.argtypes = [vp, vp, C.c_size_t]
    L.ocr_rotate_crop.argtypes = [vp, C.c_int, C.c_int, C.c_size_t, vp, C.c_int, vp, C.c_size_t, vp, ip, ip]
    L.ocr_rotate_crop_shape.argtypes = [C.c_int, C.c_int, vp, ip, ip]
    L.ocr_rotate180_rois.argtypes = [vp, C.c_int, C.c_int, C.c_size_t, vp, C.c_int]
    L._pipe_protos_done = True


CROP_BOUNDING_RECT, CROP_ROTATE = 0, 1


def rotate_crop_shape(rows, cols, box):
    """ocr_rotate_crop_shape: (rows, cols) of GetRotateCropImage's result (host arithmetic only)."""
    L = lib()
    _pipe_protos(L)
    b = np.ascontiguousarray(np.asarray(box, np.int32).reshape(8))
    r, c = C.c_int(), C.c_int()
    check(L.ocr_rotate_crop_shape(rows, cols, b.ctypes.data, C.byref(r), C.byref(c)))
    return r.value, c.value


def rotate180_rois(img, rects):
    """ocr_rotate180_rois: in-place 180-degree rotation of the (x, y, w, h) rectangles of a BGR u8 image, in list order."""
    L = lib()
    _pipe_protos(L)
    img = np.ascontiguousarray(img, dtype=np.uint8).copy()
    r = np.ascontiguousarray(rects, dtype=np.int32).reshape(-1, 4)
    check(L.ocr_rotate180_rois(img.ctypes.data, img.shape[0], img.shape[1], img.strides[0], r.ctypes.data, len(r)))
    return img


def rotate_crops(img, boxes):
    """Utility::GetRotateCropImage for each box of one image on the device (ocr_rotate_crop)."""
    L = lib()
    _pipe_protos(L)
    img = np.asarray(img, np.uint8)
    assert img.ndim == 3 and img.shape[2] == 3 and img.strides[1] == 3 and img.strides[2] == 1
    b = np.ascontiguousarray(np.asarray(boxes, np.int32).reshape(-1, 8))
    n = b.shape[0]
    shapes = [rotate_crop_shape(img.shape[0], img.shape[1], b[k]) for k in range(n)]
    cap = sum(r * c * 3 for r, c in shapes)
    out = np.empty(max(cap, 1), np.uint8)
    off = np.zeros(n + 1, np.uint64)
    rr, cc = (C.c_int * n)(), (C.c_int * n)()
    check(L.ocr_rotate_crop(img.ctypes.data, img.shape[0], img.shape[1], img.strides[0], b.ctypes.data, n, out.ctypes.data,
                            cap, off.ctypes.data, rr, cc))
    return [out[int(off[k]):int(off[k + 1])].reshape(rr[k], cc[k], 3).copy() for k in range(n)]


class DevArray:
    """A device allocation filled from a numpy array (inputs 'already resident in HBM')."""

    def __init__(self, host):
        L = lib()
        _pipe_protos(L)
        host = np.ascontiguousarray(host)
        self.nbytes = host.nbytes
        self.shape, self.dtype = host.shape, host.dtype
        self.ptr = C.c_void_p()
        check(L.ocr_dev_alloc(C.byref(self.ptr), self.nbytes))
        check(L.ocr_dev_upload(self.ptr, host.ctypes.data, self.nbytes))

    def download(self):
        out = np.empty(self.shape, self.dtype)
        check(lib().ocr_dev_download(out.ctypes.data, self.ptr, self.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().ocr_dev_free(self.ptr)
            self.ptr = None


class Pipe:
    """OCRWorker::processRequest over the C-ABI (ocr_pipe_*) for batches of images."""

    def __init__(self, model_root=None, device=0, enable_cls=False, limit_type="max", limit_side_len=512, thresh=0.2,
                 box_thresh=0.4, unclip_ratio=1.8, use_dilation=False, rec_batch_num=16, rec_img_h=28, rec_img_w=192,
                 cls_batch_num=8, crop_mode=CROP_BOUNDING_RECT, score_mode="fast", rec_sort_mode=0, phases=0, cv_compat=0,
                 precision="fp32", det_dir=None, rec_dir=None):
        """det_dir / rec_dir: model directories of their own (BASELINE configs[4]: models_server/det, models_server/rec -
        directories whose arch.txt names a server plan); the dictionary stays the reference's"""
        L = lib()
        _pipe_protos(L)
        root = model_root or MODELS
        cfg = ocr_pipe_cfg()
        L.ocr_pipe_cfg_default(C.byref(cfg))
        self._keep = [(det_dir or os.path.join(root, "det")).encode(), os.path.join(root, "cls").encode(),
                      (rec_dir or os.path.join(root, "rec")).encode(), os.path.join(root, "rec", "ppocr_keys_v1.txt").encode(),
                      limit_type.encode(), score_mode.encode(), precision.encode()]
        (cfg.det.model_dir, cfg.cls.model_dir, cfg.rec.model_dir, cfg.rec.label_path, cfg.det.limit_type,
         cfg.det.det_db_score_mode, _prec) = self._keep
        cfg.det.precision = cfg.cls.precision = cfg.rec.precision = _prec
        cfg.det.device_id = device
        cfg.det.limit_side_len = limit_side_len
        cfg.det.det_db_thresh, cfg.det.det_db_box_thresh, cfg.det.det_db_unclip_ratio = thresh, box_thresh, unclip_ratio
        cfg.det.use_dilation = int(use_dilation)
        cfg.det.cv_compat = int(cv_compat)
        cfg.rec.rec_batch_num, cfg.rec.rec_img_h, cfg.rec.rec_img_w = rec_batch_num, rec_img_h, rec_img_w
        cfg.rec.sort_mode = int(rec_sort_mode)
        cfg.cls.cls_batch_num = cls_batch_num
        cfg.enable_cls = int(enable_cls)
        cfg.crop_mode = int(crop_mode)
        cfg.phases = int(phases)
        self.h = C.c_void_p()
        check(L.ocr_pipe_create(C.byref(cfg), C.byref(self.h)))
        self.times = (C.c_double * 3)()
        self._cap_words, self._cap_ids = 0, 0

    def _bufs(self, count):
        cw, ci = count * 1000, count * 1000 * 64
        if cw > self._cap_words:
            self._words = (ocr_word * cw)()
            self._ids = np.zeros(ci, np.int32)
            self._cap_words, self._cap_ids = cw, ci
        return self._words, self._ids

    def _collect(self, count, words, ids, off, nw):
        out = []
        for i in range(count):
            ws = []
            for k in range(off[i], off[i] + nw[i]):
                w = words[k]
                ws.append(dict(box=np.array(w.box[:], np.int32).reshape(4, 2),
                               ids=ids[w.ids_off:w.ids_off + w.ids_len].copy(), confidence=float(w.confidence)))
            out.append(ws)
        return out

    def run(self, imgs):
        n = len(imgs)
        words, ids = self._bufs(n)
        off = np.zeros(n, np.int32)
        nw = np.zeros(n, np.int32)
        check(lib().ocr_pipe_run(self.h, _imgs(imgs), n, words, self._cap_words, off.ctypes.data, nw.ctypes.data,
                                 ids.ctypes.data, self._cap_ids, self.times))
        return self._collect(n, words, ids, off, nw)

    def stage(self, slot, imgs, probs=None):
        """ocr_pipe_stage: host images (any sizes) -> pinned memory -> device, asynchronously.  probs: per image a
        float32 map of the detector's input resolution (det_shape), attached with ocr_pipe_slot_probs; None keeps
        whatever maps the slot has (they survive re-staging the same sizes in the same order)."""
        n = len(imgs)
        self._staged_n = getattr(self, "_staged_n", {})
        check(lib().ocr_pipe_stage(self.h, slot, _imgs(imgs), n))
        self._staged_n[slot] = n
        if probs is not None:
            arr = (C.c_void_p * n)()
            maps = [np.ascontiguousarray(p, dtype=np.float32) for p in probs]
            for i, m in enumerate(maps):
                arr[i] = m.ctypes.data
            check(lib().ocr_pipe_slot_probs(self.h, slot, arr, n))

    def stage_dev_probs(self, slot, imgs, dev_probs):
        """ocr_pipe_stage with probability maps that already sit in device memory (DevArray per image): they are copied
        device to device into the slot (a request stream's maps are uploaded once per distinct image, not per batch)."""
        n = len(imgs)
        self._staged_n = getattr(self, "_staged_n", {})
        check(lib().ocr_pipe_stage(self.h, slot, _imgs(imgs), n))
        self._staged_n[slot] = n
        arr = (C.c_void_p * n)()
        for i, d in enumerate(dev_probs):
            arr[i] = d.ptr.value
        check(lib().ocr_pipe_slot_probs(self.h, slot, arr, n))

    def stats(self):
        """{'runs', 'binds', 'graph_replays'} summed over the pipeline's networks (Net::stats)"""
        out = (C.c_longlong * 3)()
        check(lib().ocr_pipe_stats(self.h, out))
        return dict(runs=int(out[0]), binds=int(out[1]), graph_replays=int(out[2]))

    def run_staged(self, slot, collect=True):
        count = self._staged_n[slot]
        words, ids = self._bufs(count)
        off = np.zeros(count, np.int32)
        nw = np.zeros(count, np.int32)
        check(lib().ocr_pipe_run_staged(self.h, slot, words, self._cap_words, off.ctypes.data, nw.ctypes.data, ids.ctypes.data,
                                        self._cap_ids, self.times))
        return self._collect(count, words, ids, off, nw) if collect else int(nw.sum())

    def run_device(self, dev_imgs, rows, cols, count, dev_prob=None, collect=True):
        words, ids = self._bufs(count)
        off = np.zeros(count, np.int32)
        nw = np.zeros(count, np.int32)
        check(lib().ocr_pipe_run_device(self.h, dev_imgs.ptr, rows, cols, count, dev_prob.ptr if dev_prob else None,
                                        words, self._cap_words, off.ctypes.data, nw.ctypes.data, ids.ctypes.data,
                                        self._cap_ids, self.times))
        return self._collect(count, words, ids, off, nw) if collect else int(nw.sum())

    # ---- two batches in flight (ocr_pipe_run_device_on / ocr_pipe_run_staged_on): the whole batch on one chain; calls on
    # different chains may run concurrently from different threads - result buffers and stage times are per chain
    def _chain_bufs(self, chain, count):
        if not hasattr(self, "_cb"):
            self._cb = {}
        cw, ci = count * 1000, count * 1000 * 64
        b = self._cb.get(chain)
        if b is None or b[2] < cw:
            b = self._cb[chain] = ((ocr_word * cw)(), np.zeros(ci, np.int32), cw, ci, (C.c_double * 3)())
        return b

    def run_device_on(self, chain, dev_imgs, rows, cols, count, dev_prob=None, collect=True):
        words, ids, cw, ci, times = self._chain_bufs(chain, count)
        off = np.zeros(count, np.int32)
        nw = np.zeros(count, np.int32)
        check(lib().ocr_pipe_run_device_on(self.h, chain, dev_imgs.ptr, rows, cols, count, dev_prob.ptr if dev_prob else None,
                                           words, cw, off.ctypes.data, nw.ctypes.data, ids.ctypes.data, ci, times))
        return self._collect(count, words, ids, off, nw) if collect else int(nw.sum())

    def run_staged_on(self, chain, slot, collect=True):
        count = self._staged_n[slot]
        words, ids, cw, ci, times = self._chain_bufs(chain, count)
        off = np.zeros(count, np.int32)
        nw = np.zeros(count, np.int32)
        check(lib().ocr_pipe_run_staged_on(self.h, chain, slot, words, cw, off.ctypes.data, nw.ctypes.data, ids.ctypes.data, ci, times))
        return self._collect(count, words, ids, off, nw) if collect else int(nw.sum())

    def chain_times(self, chain):
        return list(self._cb[chain][4]) if hasattr(self, "_cb") and chain in self._cb else None

    def det_shape(self, rows, cols):
        a, b = C.c_int(), C.c_int()
        check(lib().ocr_pipe_det_shape(self.h, rows, cols, C.byref(a), C.byref(b)))
        return a.value, b.value

    def label(self, i):
        s = lib().ocr_pipe_label(self.h, int(i))
        return s.decode("utf-8") if s is not None else None

    def timing(self, on=True, only=None):
        """HIP events around network launches; `only`: restrict them to launches whose name contains it."""
        check(lib().ocr_pipe_timing_filter(self.h, only.encode() if only else None))
        check(lib().ocr_pipe_timing(self.h, int(on)))

    def timing_report(self):
        buf = C.create_string_buffer(1 << 24)   # one row per (launch, bound shape): a mixed-size batch has tens of thousands
        check(lib().ocr_pipe_timing_report(self.h, buf, len(buf)))
        out = {}
        for line in buf.value.decode().splitlines():
            name, ms, cnt, fl, by = line.split()
            r = out.setdefault(name, dict(ms=0.0, count=0, flops=0.0, bytes=0.0))  # both rec lanes share names
            r["ms"] += float(ms)
            r["count"] += int(cnt)
            r["flops"] += float(fl)
            r["bytes"] += float(by)
        return out

    def close(self):
        if self.h:
            lib().ocr_pipe_destroy(self.h)
            self.h = None
