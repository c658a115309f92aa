"""Developer script (not a pytest file): layer-by-layer HIP vs oracle comparison."""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
spec = importlib.util.spec_from_file_location("cpp_paddle_ocr_amd", os.path.join(ROOT, "cpp-paddle-ocr_amd", "__init__.py"),
                                              submodule_search_locations=[os.path.join(ROOT, "cpp-paddle-ocr_amd")])
pkg = importlib.util.module_from_spec(spec)
sys.modules["cpp_paddle_ocr_amd"] = pkg
spec.loader.exec_module(pkg)
from oracle import OracleNet, lib as olib  # noqa: E402

rs = np.random.RandomState(0)
a = np.concatenate([rs.randn(4096) * 10, rs.rand(4096) * 100 - 50, [0.0, -0.0, 1.0, 88.5, -90.0, 1e-30]]).astype(np.float32)
b = (rs.randn(a.size) * 3 + 0.01).astype(np.float32)
out = pkg.probe(a, b)
ref_exp = np.array([olib().oracle_expf(float(v)) for v in a], dtype=np.float32)
print("probe div exact:", np.array_equal(out[0], a / b), "sqrt exact:", np.array_equal(out[1], np.sqrt(np.abs(a))),
      "expf exact:", np.array_equal(out[2], ref_exp), "max exp diff", np.abs(out[2] - ref_exp).max())
fma_ref = (a.astype(np.float64) * b.astype(np.float64) + a.astype(np.float64)).astype(np.float32)
print("fma exact:", np.array_equal(out[3], fma_ref), "mul+add not contracted:", np.array_equal(out[4], (a * b) + a),
      "rint:", np.array_equal(out[5], np.rint(a * np.float32(1.44269504088896341))))

for kind, shape in (("cls", (3, 48, 192)), ("det", (2, 64, 96)), ("det", (1, 192, 384)), ("rec", (3, 48, 160)), ("rec", (2, 28, 192))):
    x = rs.randn(shape[0], shape[1], shape[2], 3).astype(np.float32)
    o = OracleNet(kind)
    yo = o.run(x)
    g = pkg.Net(kind)
    yg = g.forward(x, keep_all=True)
    print(kind, shape, "output equal:", np.array_equal(yo, yg), "maxabs", np.abs(yo - yg).max())
    bad = 0
    for tid in range(1, g.num_tensors()):
        try:
            to = o.tensor(tid)
        except Exception:
            continue
        if to.size == 0:
            continue
        tg = g.fetch(tid)
        if to.shape != tg.shape:
            print("  tid", tid, "shape mismatch", to.shape, tg.shape)
            bad += 1
            continue
        if not np.array_equal(to, tg):
            d = np.abs(to - tg)
            print("  tid %d shape %s mismatch: max %.3e  frac %.4f nan %d" % (tid, to.shape, np.nanmax(d), (d > 0).mean(), np.isnan(tg).sum()))
            bad += 1
            if bad > 6:
                break
    g.close()
