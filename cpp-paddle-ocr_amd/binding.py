"""ctypes binding of include/ocr_hip.h (plumbing for tests and bench.py).

Fails loudly when the HIP library is missing or no gfx950 device is visible: there is no
CPU fallback anywhere in the product path.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_PATH = os.path.join(HERE, "lib", "libocr_hip.so")
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
                ("precision", C.c_char_p), ("max_batch", C.c_int)]


class ocr_cls_cfg(C.Structure):
    _fields_ = [("model_dir", C.c_char_p), ("device_id", C.c_int), ("cls_thresh", C.c_double),
                ("cls_batch_num", C.c_int), ("precision", C.c_char_p)]


class ocr_rec_cfg(C.Structure):
    _fields_ = [("model_dir", C.c_char_p), ("device_id", C.c_int), ("label_path", C.c_char_p),
                ("rec_batch_num", C.c_int), ("rec_img_h", C.c_int), ("rec_img_w", C.c_int),
                ("precision", C.c_char_p)]


# every symbol include/ocr_hip.h declares (tests check the .so exports all of them)
EXPORTS = [
    "ocr_last_error", "ocr_rt_init", "ocr_rt_device_count",
    "ocr_det_cfg_default", "ocr_det_create", "ocr_det_destroy", "ocr_det_run", "ocr_det_run_batch",
    "ocr_det_last_shape", "ocr_det_prob_map", "ocr_det_bitmap", "ocr_det_resized", "ocr_det_post",
    "ocr_cls_cfg_default", "ocr_cls_create", "ocr_cls_destroy", "ocr_cls_run", "ocr_cls_probs",
    "ocr_rec_cfg_default", "ocr_rec_create", "ocr_rec_destroy", "ocr_rec_run", "ocr_rec_label",
    "ocr_rec_num_classes", "ocr_rec_steps",
    "ocr_net_create", "ocr_net_destroy", "ocr_net_forward", "ocr_net_num_tensors", "ocr_net_fetch",
    "ocr_net_timing", "ocr_net_timing_report", "ocr_probe",
]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OcrError("libocr_hip.so is not built (%s): run __graft_entry__.build()" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.ocr_last_error.restype = C.c_char_p
        L.ocr_net_create.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
        L.ocr_net_destroy.argtypes = [C.c_void_p]
        L.ocr_net_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ocr_net_num_tensors.argtypes = [C.c_void_p]
        L.ocr_net_fetch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]
        L.ocr_net_timing.argtypes = [C.c_void_p, C.c_int]
        L.ocr_net_timing_report.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.ocr_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise OcrError("libocr_hip error %d: %s" % (rc, lib().ocr_last_error().decode(errors="replace")))


def probe(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    out = np.empty((6, a.size), dtype=np.float32)
    check(lib().ocr_probe(a.ctypes.data, b.ctypes.data, out.ctypes.data, a.size))
    return out


class Net:
    """Raw network tap (ocr_net_*): host f32 NHWC in, logical NHWC tensors out."""

    def __init__(self, kind, model_dir=None, weights=None, device=0):
        self.kind = kind
        model_dir = model_dir or os.path.join(MODELS, kind)
        self.h = C.c_void_p()
        check(lib().ocr_net_create(kind.encode(), model_dir.encode(), weights.encode() if weights else None, device,
                                   C.byref(self.h)))

    def forward(self, x_nhwc, keep_all=False):
        x = np.ascontiguousarray(x_nhwc, dtype=np.float32)
        n, h, w, c = x.shape
        assert c == 3
        check(lib().ocr_net_forward(self.h, x.ctypes.data, n, h, w, int(keep_all)))
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
