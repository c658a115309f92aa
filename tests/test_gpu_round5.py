"""Round-5 GPU tests (through the C-ABI): the bind-time probe of the folded concat, the LAB fold's stage lists as the device
executes them, the OpenCV-rule blast radius on the benchmark batch, the worker pool on two logical devices."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_concat_fold_refused_at_bind_keeps_the_materialised_concat(pkg, built):
    """the DB neck's concat is folded into the 3x3 conv's tile fill only when that kernel's launcher
    accepts the shape AT BIND TIME (shape checks + the device's dynamic-LDS attribute).  Refused there - injected here as a
    failed attribute would refuse it - the concat stays a launch of its own and the conv keeps its fallback chain: the run
    succeeds and the output is the oracle's."""
    from oracle import OracleNet
    L = pkg.lib()
    x = np.random.RandomState(5).randn(2, 96, 160, 3).astype(np.float32)
    want = OracleNet("det").run(x)
    names = {}
    for refuse in (None, b"conv3x3_c24@bind"):
        assert L.ocr_selftest_refuse_launch(refuse) == 0
        try:
            g = pkg.Net("det")
            g.timing(True)
            got = g.forward(x, keep_all=False)
            names[refuse] = sorted(g.timing_report())
            g.close()
        finally:
            L.ocr_selftest_refuse_launch(None)
        assert np.array_equal(want, got), refuse
    assert any("_cat4" in n for n in names[None]) and not any(".concat_" in n for n in names[None])
    assert any(".concat_" in n for n in names[b"conv3x3_c24@bind"]) and not any("_cat4" in n for n in names[b"conv3x3_c24@bind"])


def test_lab_fold_stage_lists_on_the_device(pkg, built):
    """The round-5 arithmetic contract: the loader folds `bias | *s0 | +a0 [| hswish | *s1 | +a1]` into the
    weights, one bias and one fma; a depthwise conv whose only reader is a 1x1 conv hands its scale / shift to that conv.
    Every materialised tensor of the unfused launch list (keep_all = 1: the absorbed depthwise tensors hold the bare
    hard-swish product there) equals the oracle's, which folds with the same operations."""
    from oracle import OracleNet
    for kind, shape in (("rec", (2, 48, 96)), ("det", (1, 64, 96))):
        x = np.random.RandomState(11).randn(shape[0], shape[1], shape[2], 3).astype(np.float32)
        o, g = OracleNet(kind), pkg.Net(kind)
        want = o.run(x)
        got = g.forward(x, keep_all=True)
        assert np.array_equal(want, got)
        checked = 0
        for tid in range(1, g.num_tensors()):
            if not g.exists(tid):
                continue
            a, b = g.fetch(tid), o.tensor(tid)
            assert a.size == b.size and np.array_equal(a.reshape(-1), b.reshape(-1)), (kind, tid)
            checked += 1
        assert checked >= 50
        g.close()


def _word_key(w):
    return (tuple(np.asarray(w["box"]).reshape(-1).tolist()), tuple(np.asarray(w["ids"]).tolist()))


def test_cv_compat_blast_radius_through_the_full_pipeline(pkg, built, card):
    """what a wrong `cv_compat` default would cost.  The SAME inputs run through the full pipeline
    (det + cls + rec) under OpenCV's 4.5.1 fillPoly rule (45) and the 4.5.2+ rule (410, the default): every run equals the
    oracle under ITS rule on sampled images, and the number of words that differ between the two rules is recorded
    (gpurun_out/r5_cv_compat_blast_radius.json when that directory is writable) - for the benchmark batch (configs[1]: 64
    images whose probability maps are the protocol's clean rectangles), for mixed-size configs[2] samples, and for the
    reference's own test image through the detector network (synthetic weights: a noisy map with boxes near box_thresh)."""
    import json
    from pipeline import Pipeline, DetCfg
    from synth_data import cfg2_sample, cfg3_item
    kw = dict(rec_batch_num=16, rec_img_h=48, rec_img_w=320, enable_cls=True)
    s2 = [cfg2_sample(i) for i in range(64)]
    s3 = [cfg3_item(i) for i in range(8)]
    report = {}
    runs = {}
    for compat in (45, 410):
        pg = pkg.Pipe(limit_side_len=960, cv_compat=compat, **kw)
        po = Pipeline(det_cfg=DetCfg(limit_side_len=960, cv_compat=compat), **kw)
        pg.stage(0, [s[0] for s in s2], [s[1] for s in s2])
        a = pg.run_staged(0)
        pg.stage(1, [s[0] for s in s3], [s[1] for s in s3])
        b = pg.run_staged(1)
        c = pg.run([card])
        for got, (img, prob) in ((a[5], s2[5][:2]), (a[40], s2[40][:2]), (b[2], s3[2][:2])):
            w = po.process(img, prob)["words"]
            assert [_word_key(x) for x in got] == [_word_key(x) for x in w], compat
        w = po.process(card)["words"]
        assert [_word_key(x) for x in c[0]] == [_word_key(x) for x in w], compat
        runs[compat] = {"configs1_64_images": a, "configs2_8_images": b, "reference_card_through_det_net": c}
        pg.close()
    for name in runs[45]:
        x, y = runs[45][name], runs[410][name]
        n45 = sum(len(i) for i in x)
        n410 = sum(len(i) for i in y)
        diff = sum(len(set(map(_word_key, i)) ^ set(map(_word_key, j))) for i, j in zip(x, y))
        report[name] = {"words_under_45": n45, "words_under_410": n410, "words_in_one_result_only": diff}
    assert report["configs1_64_images"]["words_under_410"] == 64 * 32
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out) and os.access(out, os.W_OK):
        json.dump(report, open(os.path.join(out, "r5_cv_compat_blast_radius.json"), "w"), indent=1)
    print("cv_compat blast radius:", json.dumps(report))


def test_strict_msvc_sort_mode_refuses_what_is_not_restated(pkg, built):
    """MSVC's std::sort keeps ties in input order only up to 32 elements.  OCR_SORT_STABLE keeps input
    order beyond that too (documented: not MSVC's order there); OCR_SORT_MSVC_STRICT refuses an image with more than 32
    crops of which two have equal ratios - an error with a message, not a silently different batch composition - and
    equals OCR_SORT_STABLE wherever MSVC's order IS restated (up to 32 crops, or no ties)."""
    rs = np.random.RandomState(32)
    tied40 = [rs.randint(0, 255, [(24, 96), (12, 48), (30, 200)][i % 3] + (3,)).astype(np.uint8) for i in range(40)]
    untied40 = [rs.randint(0, 255, (24, 60 + 3 * i, 3)).astype(np.uint8) for i in range(40)]
    kw = dict(rec_batch_num=16, rec_img_h=48, rec_img_w=320)
    strict, stable = pkg.Rec(sort_mode=2, **kw), pkg.Rec(sort_mode=1, **kw)
    with pytest.raises(pkg.OcrError, match="MSVC"):
        strict.run(tied40)
    for crops in (tied40[:32], untied40):
        (ta, sa), (tb, sb) = strict.run(crops), stable.run(crops)
        assert all(np.array_equal(x, y) for x, y in zip(ta, tb)) and list(sa) == list(sb)
    strict.close()
    stable.close()


def test_two_batches_in_flight_on_one_handle(pkg, built):
    """ocr_pipe_run_device_on / ocr_pipe_run_staged_on: the whole batch on ONE chain, two chains driven concurrently by two host
    threads with DIFFERENT batches (consecutive requests overlapping, the worker pool's shape on one handle).  Every batch's
    words equal what the plain call returns for it - and the plain call equals the oracle elsewhere in this suite."""
    import threading
    from synth_data import cfg2_sample, cfg3_item
    kw = dict(rec_batch_num=16, rec_img_h=48, rec_img_w=320, enable_cls=True, limit_side_len=960)
    a = [cfg2_sample(i) for i in range(6)]
    b = [cfg3_item(i) for i in range(5)]
    p = pkg.Pipe(**kw)
    p.stage(0, [s[0] for s in a], [s[1] for s in a])
    want_a = p.run_staged(0)
    p.stage(1, [s[0] for s in b], [s[1] for s in b])
    want_b = p.run_staged(1)
    got, errs = {}, []

    def work(chain, slot, reps):
        try:
            for _ in range(reps):
                got[chain] = p.run_staged_on(chain, slot)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=work, args=(0, 0, 4)), threading.Thread(target=work, args=(1, 1, 5))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    for g, w in ((got[0], want_a), (got[1], want_b)):
        assert len(g) == len(w)
        for x, y in zip(g, w):
            assert [_word_key(i) for i in x] == [_word_key(i) for i in y]
    # resident inputs, the benchmark's call
    imgs, probs = np.stack([s[0] for s in a]), np.stack([s[1] for s in a])
    d_i, d_p = pkg.DevArray(imgs), pkg.DevArray(probs)
    plain = p.run_device(d_i, 960, 960, 6, d_p)
    for chain in (0, 1):
        on = p.run_device_on(chain, d_i, 960, 960, 6, d_p)
        assert [[_word_key(i) for i in x] for x in on] == [[_word_key(i) for i in x] for x in plain]
    with pytest.raises(pkg.OcrError):
        p.run_device_on(2, d_i, 960, 960, 6, d_p)
    d_i.free()
    d_p.free()
    p.close()
