"""TEST INFRASTRUCTURE — ctypes front-end to the CPU oracle (oracle/liboracle.so) and to the
compiled reference clipper (oracle/_ref/libclipper_ref.so).  Only tests/, smoke() and bench.py's
cpu_baseline leg may import this.  The product never does."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from pdmodel import Program, read_params  # noqa: E402

_lib = None


def usable_cores():
    """CPU threads this process may actually use (affinity mask and cgroup quota), capped at 32:
    the oracle's loops are small and oversubscribed OpenMP teams make it slower, not faster."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 32))


def lib():
    global _lib
    if _lib is None:
        os.environ.setdefault("OMP_NUM_THREADS", str(usable_cores()))
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liboracle.so missing: run `make -C oracle` (or __graft_entry__.build())")
        _lib = C.CDLL(path)
        _lib.oracle_net_create.restype = C.c_void_p
        _lib.oracle_net_create.argtypes = [C.c_char_p]
        _lib.oracle_net_destroy.argtypes = [C.c_void_p]
        _lib.oracle_net_set_param.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_void_p]
        _lib.oracle_net_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        _lib.oracle_net_error.restype = C.c_char_p
        _lib.oracle_net_error.argtypes = [C.c_void_p]
        _lib.oracle_net_tensor.restype = C.c_long
        _lib.oracle_net_tensor.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _lib.oracle_expf.restype = C.c_float
        _lib.oracle_expf.argtypes = [C.c_float]
    return _lib


def plan_text(net):
    return open(os.path.join(ROOT, "cpp-paddle-ocr_amd", "plans", net + ".plan")).read()


def load_weights(net, model_root=None):
    """real inference.pdiparams if present, else the seeded synthetic file."""
    if net.startswith("srv_"):  # BASELINE configs[4]: hand-written plans, seeded weights only (tools/make_server_plans.py)
        import synth_weights
        kind = net[4:]
        path = synth_weights.ensure_server(ROOT)[("det", "rec").index(kind)]
        names = [n for n, _ in synth_weights.server_param_table(os.path.join(ROOT, "cpp-paddle-ocr_amd", "plans", net + ".plan"))]
        return read_params(path, sorted(names))
    model_root = model_root or os.path.join(ROOT, "models")
    d = os.path.join(model_root, net)
    prog = Program(os.path.join(d, "inference.pdmodel"))
    for fn in ("inference.pdiparams", "synthetic.pdiparams"):
        p = os.path.join(d, fn)
        if os.path.exists(p):
            return read_params(p, prog.persistable_names())
    import synth_weights
    synth_weights.ensure(os.path.dirname(model_root))
    return read_params(os.path.join(d, "synthetic.pdiparams"), prog.persistable_names())


class OracleNet:
    def __init__(self, net, weights=None, plan=None):
        """plan: another plan text on the net's weights (tests: the detector's plan without its final sigmoid, to read logits)"""
        self.net = net
        L = lib()
        self.h = L.oracle_net_create((plan if plan is not None else plan_text(net)).encode())
        assert self.h, "plan parse failed"
        self.weights = weights if weights is not None else load_weights(net)
        for name, a in self.weights.items():
            a = np.ascontiguousarray(a, dtype=np.float32)
            dims = (C.c_int * max(1, a.ndim))(*(a.shape if a.ndim else (1,)))
            L.oracle_net_set_param(self.h, name.encode(), a.ctypes.data, max(1, a.ndim), dims)

    def run(self, x_nhwc):
        x = np.ascontiguousarray(x_nhwc, dtype=np.float32)
        n, h, w, c = x.shape
        assert c == 3
        rc = lib().oracle_net_run(self.h, x.ctypes.data, n, h, w)
        if rc:
            raise RuntimeError(lib().oracle_net_error(self.h).decode())
        return self.tensor(-1)

    def logits(self):
        """Input of the final softmax op (arg max is taken over the logits: DESIGN.md section 4)."""
        if not hasattr(self, "_logits_tid"):
            tid = None
            for line in plan_text(self.net).splitlines():
                if line.startswith("softmax "):
                    tid = int([f for f in line.split() if f.startswith("i=")][0][2:])
            self._logits_tid = tid
        return self.tensor(self._logits_tid)

    def tensor(self, tid):
        dims = (C.c_int * 4)()
        ptr = C.c_void_p()
        n = lib().oracle_net_tensor(self.h, tid, dims, C.byref(ptr))
        a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(n,)).copy()
        return a.reshape(dims[0], dims[1], dims[2], dims[3])

    def __del__(self):
        try:
            lib().oracle_net_destroy(self.h)
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------
# image / post-processing oracle functions (oracle_img.cpp, oracle_post.cpp)
# ---------------------------------------------------------------------------------------------
def _p(a):
    return a.ctypes.data


def resize_u8c3(img, dh, dw):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.empty((dh, dw, 3), np.uint8)
    lib().oracle_resize_u8c3(C.c_void_p(_p(img)), img.shape[0], img.shape[1], C.c_size_t(img.strides[0]),
                             C.c_void_p(_p(out)), dh, dw)
    return out


def det_resize_shape(h, w, limit_type="max", limit_side_len=512):
    rh, rw = C.c_int(), C.c_int()
    fh, fw = C.c_float(), C.c_float()
    lib().oracle_det_resize_shape(h, w, limit_type.encode(), limit_side_len, C.byref(rh), C.byref(rw), C.byref(fh),
                                  C.byref(fw))
    return rh.value, rw.value, fh.value, fw.value


def det_preprocess(img, rh, rw):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.empty((rh, rw, 3), np.float32)
    res = np.empty((rh, rw, 3), np.uint8)
    lib().oracle_det_preprocess(C.c_void_p(_p(img)), img.shape[0], img.shape[1], C.c_size_t(img.strides[0]), rh, rw,
                                C.c_void_p(_p(out)), C.c_void_p(_p(res)))
    return out, res


def bitmap(pred, thresh, use_dilation=False):
    pred = np.ascontiguousarray(pred, dtype=np.float32)
    bm = np.empty(pred.shape, np.uint8)
    lib().oracle_bitmap(C.c_void_p(_p(pred)), pred.shape[0], pred.shape[1], C.c_double(thresh), int(use_dilation),
                        C.c_void_p(_p(bm)))
    return bm


CV_45, CV_410 = 45, 410  # include/ocr_hip.h OCR_CV_45 / OCR_CV_410 (0 = default = CV_410)


def det_post(pred, thresh, box_thresh, unclip_ratio, src_h, src_w, use_dilation=False, slow=False, cap=2000, cv_compat=0):
    pred = np.ascontiguousarray(pred, dtype=np.float32)
    boxes = np.zeros((cap, 8), np.int32)
    n = lib().oracle_det_post(C.c_void_p(_p(pred)), pred.shape[0], pred.shape[1], C.c_double(thresh),
                              C.c_double(box_thresh), C.c_double(unclip_ratio), int(use_dilation), int(slow), src_h,
                              src_w, C.c_void_p(_p(boxes)), cap, int(cv_compat))
    assert n <= cap
    return boxes[:n].reshape(n, 4, 2).copy()


def crop_rect(box, rows, cols):
    b = np.ascontiguousarray(np.asarray(box, dtype=np.int32).reshape(8))
    x, y, w, h = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    ok = lib().oracle_crop_rect(C.c_void_p(_p(b)), rows, cols, C.byref(x), C.byref(y), C.byref(w), C.byref(h))
    return (x.value, y.value, w.value, h.value) if ok else None


def rotate_crop(img, box):
    """Utility::GetRotateCropImage (utility.cpp:137-190); None when the bounding-box crop is empty."""
    img = np.asarray(img, dtype=np.uint8)
    assert img.strides[1] == 3 and img.strides[2] == 1
    b = np.ascontiguousarray(np.asarray(box, dtype=np.int32).reshape(8))
    r, c = C.c_int(), C.c_int()
    if not lib().oracle_rotate_crop_shape(img.shape[0], img.shape[1], C.c_void_p(_p(b)), C.byref(r), C.byref(c)):
        return None
    out = np.empty((r.value, c.value, 3), np.uint8)
    lib().oracle_rotate_crop(C.c_void_p(img.ctypes.data), img.shape[0], img.shape[1], C.c_size_t(img.strides[0]),
                             C.c_void_p(_p(b)), C.c_void_p(_p(out)), C.byref(r), C.byref(c))
    return out


def rec_preprocess(crop, imgH, imgW):
    crop = np.asarray(crop, dtype=np.uint8)
    out = np.empty((imgH, imgW, 3), np.float32)
    lib().oracle_rec_preprocess(C.c_void_p(crop.ctypes.data), crop.shape[0], crop.shape[1],
                                C.c_size_t(crop.strides[0]), imgH, imgW, C.c_void_p(_p(out)))
    return out


def cls_preprocess(crop):
    crop = np.asarray(crop, dtype=np.uint8)
    out = np.empty((48, 192, 3), np.float32)
    lib().oracle_cls_preprocess(C.c_void_p(crop.ctypes.data), crop.shape[0], crop.shape[1],
                                C.c_size_t(crop.strides[0]), C.c_void_p(_p(out)))
    return out


def rotate180_inplace(view):
    assert view.dtype == np.uint8 and view.strides[1] == 3 and view.strides[2] == 1
    lib().oracle_rotate180_inplace(C.c_void_p(view.ctypes.data), view.shape[0], view.shape[1],
                                   C.c_size_t(view.strides[0]))


def argsort(v, stable=False):
    v = np.ascontiguousarray(v, dtype=np.float32)
    idx = np.empty(v.size, np.int32)
    (lib().oracle_argsort_stable if stable else lib().oracle_argsort)(C.c_void_p(_p(v)), v.size, C.c_void_p(_p(idx)))
    return idx


def ctc_decode(amax, pmax):
    amax = np.ascontiguousarray(amax, dtype=np.int32)
    pmax = np.ascontiguousarray(pmax, dtype=np.float32)
    ids = np.empty(amax.size, np.int32)
    score = C.c_float()
    n = lib().oracle_ctc_decode(C.c_void_p(_p(amax)), C.c_void_p(_p(pmax)), amax.size, C.c_void_p(_p(ids)),
                                C.byref(score))
    if n < 0:
        return None, None
    return ids[:n].copy(), score.value
