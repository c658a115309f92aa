"""Developer script: fused pipeline vs oracle pipeline (card image; cfg2 sample with prob override)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from __graft_entry__ import load_package, smoke  # noqa: E402
pkg = load_package()
import oracle as O  # noqa: E402
from pipeline import Pipeline, DetCfg  # noqa: E402
from synth_data import cfg2_sample  # noqa: E402

smoke()
card = np.load(os.path.join(ROOT, "tests", "golden", "card_jd_bgr.npy"))
for cls_on in (False, True):
    pg = pkg.Pipe(enable_cls=cls_on)
    po = Pipeline(enable_cls=cls_on)
    t = time.time()
    got = pg.run([card, card[:, ::-1].copy(), card])
    print("pipe run 3 imgs %.1f ms" % ((time.time() - t) * 1e3), list(pg.times))
    want = [po.process(card), po.process(card[:, ::-1].copy()), po.process(card)]
    for i in range(3):
        g, w = got[i], want[i]["words"]
        ok = len(g) == len(w) and all(np.array_equal(a["box"], b["box"]) and np.array_equal(a["ids"], b["ids"]) and
                                      a["confidence"] == np.float32(b["confidence"]) for a, b in zip(g, w))
        print(" cls", cls_on, "img", i, len(g), len(w), "IDENTICAL" if ok else "MISMATCH")
    pg.close()

# bench protocol on 2 images
imgs, probs = [], []
for i in range(2):
    a, b, _ = cfg2_sample(i)
    imgs.append(a)
    probs.append(b)
imgs, probs = np.stack(imgs), np.stack(probs)
pg = pkg.Pipe(enable_cls=True, limit_side_len=960, rec_batch_num=16, rec_img_h=48, rec_img_w=320)
d_i, d_p = pkg.DevArray(imgs), pkg.DevArray(probs)
got = pg.run_device(d_i, 960, 960, 2, d_p)
t = time.time()
got = pg.run_device(d_i, 960, 960, 2, d_p)
print("run_device 2 imgs %.1f ms" % ((time.time() - t) * 1e3), list(pg.times))
po = Pipeline(det_cfg=DetCfg(limit_side_len=960), rec_batch_num=16, rec_img_h=48, rec_img_w=320, enable_cls=True)
for i in range(2):
    img = imgs[i].copy()
    boxes = O.det_post(probs[i], 0.2, 0.4, 1.8, 960, 960)
    views = []
    for b in boxes:
        r = O.crop_rect(b, 960, 960)
        if r:
            x, y, w, h = r
            views.append(img[y:y + h, x:x + w])
    labels, _ = po.cls_run(views)
    for k, v in enumerate(views):
        if labels[k] == 1:
            O.rotate180_inplace(v)
    texts, scores, _ = po.rec_run(views)
    g = got[i]
    ok = len(g) == len(texts) and all(np.array_equal(g[k]["box"], boxes[k]) and np.array_equal(g[k]["ids"], texts[k]) and
                                      g[k]["confidence"] == scores[k] for k in range(len(g)))
    print(" cfg2 img", i, len(g), len(texts), "rot", int(labels.sum()), "IDENTICAL" if ok else "MISMATCH")
