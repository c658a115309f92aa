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


def lib():
    global _lib
    if _lib is None:
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
    def __init__(self, net, weights=None):
        L = lib()
        self.h = L.oracle_net_create(plan_text(net).encode())
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
