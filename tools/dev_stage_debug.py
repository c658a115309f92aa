"""Developer script (not pytest): stage-level HIP vs oracle comparison with diagnostics."""
import importlib.util
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
spec = importlib.util.spec_from_file_location("cpp_paddle_ocr_amd", os.path.join(ROOT, "cpp-paddle-ocr_amd", "__init__.py"),
                                              submodule_search_locations=[os.path.join(ROOT, "cpp-paddle-ocr_amd")])
pkg = importlib.util.module_from_spec(spec)
sys.modules["cpp_paddle_ocr_amd"] = pkg
spec.loader.exec_module(pkg)
import oracle as O  # noqa: E402
from pipeline import Pipeline, DetCfg  # noqa: E402
from synth_data import cfg2_sample, make_layout, make_image, make_prob_map  # noqa: E402

card = np.load(os.path.join(ROOT, "tests", "golden", "card_jd_bgr.npy"))


def cmp_boxes(name, a, b):
    ok = len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b))
    print("  %s: %d vs %d boxes -> %s" % (name, len(a), len(b), "IDENTICAL" if ok else "MISMATCH"))
    if not ok:
        sa = {tuple(np.asarray(x).ravel()) for x in a}
        sb = {tuple(np.asarray(x).ravel()) for x in b}
        print("    only oracle:", sorted(sa - sb)[:4], " only hip:", sorted(sb - sa)[:4], "set-equal:", sa == sb)
    return ok


# ---- det on the card image (worker defaults) ----
pipe = Pipeline()
det = pkg.Det()
bo = pipe.det_run(card)
t = time.time()
bg = det.run(card)
print("det card: times", list(det.times), "wall %.1f ms" % ((time.time() - t) * 1e3))
print("  resized equal:", np.array_equal(det.resized(), pipe.taps["det_resized"]),
      " prob equal:", np.array_equal(det.prob_map(), pipe.taps["det_prob"]),
      " bitmap equal:", np.array_equal(det.bitmap(), pipe.taps["det_bitmap"]), "fg px", int(pipe.taps["det_bitmap"].sum()))
cmp_boxes("card boxes", bo, bg)

# ---- det post on synthetic probability maps (cfg2 protocol) ----
det960 = pkg.Det(limit_side_len=960, thresh=0.3, box_thresh=0.5, unclip_ratio=2.0)
allok = True
for i in range(6):
    img, prob, layout = cfg2_sample(i)
    t0 = time.time()
    bo = O.det_post(prob, 0.3, 0.5, 2.0, 960, 960)
    t1 = time.time()
    bg = det960.post(prob, 960, 960)
    t2 = time.time()
    allok &= cmp_boxes("synth map %d (oracle %.1f ms, hip %.1f ms)" % (i, (t1 - t0) * 1e3, (t2 - t1) * 1e3), bo, bg)
# noisy / adversarial maps: many specks and holes
rs = np.random.RandomState(3)
from scipy import ndimage
for i in range(4):
    H, W = 256 + 32 * i, 320
    f = ndimage.gaussian_filter(rs.rand(H, W), 1.5 + i)
    f = (f - f.min()) / (f.max() - f.min())
    f = f ** (2.5 + i)
    bo = O.det_post(f.astype(np.float32), 0.3, 0.5, 2.0, H, W)
    bg = det960.post(f.astype(np.float32), H, W)
    allok &= cmp_boxes("blob map %d" % i, bo, bg)
print("post all identical:", allok)

# ---- rec / cls on crops of a synthetic image ----
img, prob, layout = cfg2_sample(0)
boxes = O.det_post(prob, 0.3, 0.5, 2.0, 960, 960)
crops = []
for b in boxes:
    r = O.crop_rect(b, 960, 960)
    if r:
        x, y, w, h = r
        crops.append(img[y:y + h, x:x + w])
print("crops", len(crops), [c.shape[:2] for c in crops[:5]])
for (hh, ww, bn) in ((48, 320, 6), (28, 192, 16)):
    p2 = Pipeline(rec_batch_num=bn, rec_img_h=hh, rec_img_w=ww)
    rec = pkg.Rec(rec_batch_num=bn, rec_img_h=hh, rec_img_w=ww)
    to, so, steps = p2.rec_run(crops)
    t = time.time()
    tg, sg = rec.run(crops)
    wall = (time.time() - t) * 1e3
    ids_ok = all(np.array_equal(a, b) for a, b in zip(to, tg))
    print("rec %dx%d: ids identical %s, score maxabs %.3e (exact %s), wall %.1f ms, times %s" % (
        hh, ww, ids_ok, np.abs(so - sg).max(), np.array_equal(so, sg), wall, list(rec.times)))
    bad = 0
    for i in range(len(crops)):
        am, pm = rec.steps(i)
        if not (np.array_equal(am, steps[i][0]) and np.array_equal(pm, steps[i][1])):
            bad += 1
            if bad < 3:
                print("   line", i, "T", len(am), len(steps[i][0]), "amax eq", np.array_equal(am, steps[i][0]),
                      "pmax maxabs", np.abs(pm - steps[i][1]).max() if len(pm) == len(steps[i][1]) else None)
    print("   per-step taps mismatching lines:", bad, " sample text len", len(tg[0]), repr(rec.text(tg[0])[:20]))
    rec.close()
pc = Pipeline(enable_cls=True)
cls = pkg.Cls()
lo, so = pc.cls_run(crops)
lg, sg = cls.run(crops)
print("cls: labels identical", np.array_equal(lo, lg), "scores exact", np.array_equal(so, sg), "probs exact",
      np.array_equal(cls.probs(len(crops)), pc.taps["cls_probs"]), "labels", lg[:10])

# ---- det batch on 960 images, full network (synthetic weights) ----
imgs = [cfg2_sample(i)[0] for i in range(2)]
pipe960 = Pipeline(det_cfg=DetCfg(limit_side_len=960, thresh=0.3, box_thresh=0.5, unclip_ratio=2.0))
t = time.time()
bg = det960.run_batch(imgs)
print("det960 batch2 wall %.1f ms times %s" % ((time.time() - t) * 1e3, list(det960.times)))
t = time.time()
bo0 = pipe960.det_run(imgs[0])
print("oracle det960 %.1f s" % (time.time() - t))
print("  prob equal:", np.array_equal(det960.prob_map(0), pipe960.taps["det_prob"]), "bitmap fg", int(pipe960.taps["det_bitmap"].sum()))
cmp_boxes("det960 img0 boxes (synthetic weights)", bo0, bg[0])
