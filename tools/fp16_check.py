"""precision="fp16" against the f32 contract: per network the output / logit differences on seeded inputs, and the
pipeline's agreement with the fp32 pipeline on cfg2 images (tools/fp16_check.py [n_images])."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")]
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
rs = np.random.RandomState(0)
for kind, shape in (("cls", (4, 48, 192)), ("det", (2, 160, 224)), ("rec", (4, 48, 320))):
    x = rs.randn(*shape, 3).astype(np.float32)
    a, b = pkg.Net(kind), pkg.Net(kind, precision="fp16")
    ya, yb = a.forward(x), b.forward(x)
    d = np.abs(ya - yb)
    print("%s out %s: max|d| %.3e mean|d| %.3e  max|y| %.3e  argmax agree %.4f" % (
        kind, ya.shape, d.max(), d.mean(), np.abs(ya).max(), (ya.reshape(-1, ya.shape[-1]).argmax(1) == yb.reshape(-1, yb.shape[-1]).argmax(1)).mean()))
    a.close(); b.close()

# ---- at the LOGIT level: the softmax's input of each network, f32 contract against precision "fp16", on the
# classifier's REAL weights with real crops (the reference's test image cut into 48x192 pieces through the reference's own
# pre-processing) and on the synthetic det / rec weights.  The north star's float tolerance is 1e-3 on fp32 logits: the mode is
# outside it by construction (f16 storage rounds every activation to 11 bits), which is why it is never `value`.
def softmax_input_tid(kind):
    for line in open(os.path.join(ROOT, "cpp-paddle-ocr_amd", "plans", kind + ".plan")):
        if line.startswith("softmax "):
            return int([f for f in line.split() if f.startswith("i=")][0][2:])
    return None


import oracle as O  # noqa: E402  (test infrastructure: the reference's cls pre-processing for the crops)
card = np.load(os.path.join(ROOT, "tests", "golden", "card_jd_bgr.npy"))
crops = [card[y:y + 40, x:x + 160] for y in range(10, card.shape[0] - 40, 34) for x in range(0, card.shape[1] - 160, 110)][:48]
xc = np.stack([O.cls_preprocess(c) for c in crops])
for kind, x in (("cls", xc), ("rec", rs.randn(4, 48, 320, 3).astype(np.float32))):
    tid = softmax_input_tid(kind)
    a, b = pkg.Net(kind), pkg.Net(kind, precision="fp16")
    a.forward(x, keep_all=True); b.forward(x, keep_all=True)
    la, lb = a.fetch(tid).reshape(-1), b.fetch(tid).reshape(-1)
    d = np.abs(la - lb)
    print("%s LOGITS (%d values, %s weights): max|d| %.3e  mean|d| %.3e  max|logit| %.3e  -> max|d| / max|logit| %.3e" % (
        kind, la.size, "the reference's real" if kind == "cls" else "synthetic", d.max(), d.mean(), np.abs(la).max(), d.max() / np.abs(la).max()))
    a.close(); b.close()

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
from synth_data import cfg2_sample  # noqa: E402
imgs, probs = zip(*[cfg2_sample(i)[:2] for i in range(n)])
kw = dict(enable_cls=True, limit_side_len=960, rec_batch_num=16, rec_img_h=48, rec_img_w=320)
res = {}
for prec in ("fp32", "fp16"):
    p = pkg.Pipe(precision=prec, **kw)
    d_i, d_p = pkg.DevArray(np.stack(imgs)), pkg.DevArray(np.stack(probs))
    p.run_device(d_i, 960, 960, n, d_p, collect=False)
    t0 = time.perf_counter()
    for _ in range(3):
        w = p.run_device(d_i, 960, 960, n, d_p, collect=True)
    dt = (time.perf_counter() - t0) / 3
    res[prec] = w
    print(prec, "%.1f img/s" % (n / dt), "stage ms", list(p.times))
    p.close()
same_ids = tot = 0
conf = []
for wa, wb in zip(res["fp32"], res["fp16"]):
    assert len(wa) == len(wb)
    for a, b in zip(wa, wb):
        assert np.array_equal(a["box"], b["box"])
        tot += 1
        same_ids += np.array_equal(a["ids"], b["ids"])
        conf.append(abs(a["confidence"] - b["confidence"]))
print("words %d, identical id sequences %.4f, max |dconf| %.3e" % (tot, same_ids / tot, max(conf)))
