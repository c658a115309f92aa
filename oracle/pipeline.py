"""TEST INFRASTRUCTURE — the reference pipeline restated on top of the C oracle.

Follows, call for call:
  DBDetector::Run        /root/reference/src/ocr_det.cpp:93-176
  Classifier::Run        /root/reference/src/ocr_cls.cpp:23-106
  CRNNRecognizer::Run    /root/reference/src/ocr_rec.cpp:24-135
  OCRWorker::processRequest   /root/reference/src/ocr_worker.cpp:213-311
Only tests/, smoke() and bench.py's cpu_baseline leg may import this.
"""
import math

import numpy as np

import oracle as O


class DetCfg:
    # literals of OCRWorker::OCRWorker (ocr_worker.cpp:21-35)
    def __init__(self, limit_type="max", limit_side_len=512, thresh=0.2, box_thresh=0.4, unclip_ratio=1.8,
                 score_mode="fast", use_dilation=False, cv_compat=0):
        self.limit_type, self.limit_side_len = limit_type, limit_side_len
        self.thresh, self.box_thresh, self.unclip_ratio = thresh, box_thresh, unclip_ratio
        self.score_mode, self.use_dilation = score_mode, use_dilation
        self.cv_compat = cv_compat  # 0 / 410: OpenCV >= 4.5.2 fillPoly rule (default), 45: the 4.5.1 rule


class Pipeline:
    def __init__(self, det_cfg=None, rec_batch_num=16, rec_img_h=28, rec_img_w=192, cls_batch_num=8, enable_cls=False,
                 crop_mode="rect", rec_sort="std", det_net="det", rec_net="rec"):
        # det_net / rec_net = "srv_det" / "srv_rec": BASELINE configs[4]'s hand-written server plans (NOT reference artifacts).
        # The server recognizer (SVTR-large) has ONE token grid: every line goes into rec_img_h x rec_img_w (the batch's width
        # ratio is held at rec_img_w / rec_img_h), and its plan ends at the CTC logits: the softmax is taken here.
        self.srv_rec = rec_net.startswith("srv_")
        self.rec_sort = rec_sort    # "std": this host's std::sort; "stable": ties in input order (MSVC's std::sort up to 32 crops)
        self.crop_mode = crop_mode  # "rect": worker's ROI views; "rotate": Utility::GetRotateCropImage per box
        self.det_cfg = det_cfg or DetCfg()
        self.det = O.OracleNet(det_net)
        self.rec = O.OracleNet(rec_net)
        self.cls = O.OracleNet("cls") if enable_cls else None
        self.rec_batch_num, self.rec_img_h, self.rec_img_w = rec_batch_num, rec_img_h, rec_img_w
        self.cls_batch_num = cls_batch_num
        self.enable_cls = enable_cls
        self.taps = {}

    # ---- DBDetector::Run
    def det_run(self, img, prob_override=None):
        c = self.det_cfg
        h, w = img.shape[:2]
        rh, rw, ratio_h, ratio_w = O.det_resize_shape(h, w, c.limit_type, c.limit_side_len)
        x, resized = O.det_preprocess(img, rh, rw)
        if prob_override is None:
            prob = self.det.run(x[None])[0, :, :, 0]
        else:
            prob = prob_override
        boxes = O.det_post(prob, c.thresh, c.box_thresh, c.unclip_ratio, h, w, c.use_dilation, c.score_mode == "slow",
                           cv_compat=c.cv_compat)
        self.taps.update(det_x=x, det_resized=resized, det_prob=prob,
                         det_bitmap=O.bitmap(prob, c.thresh, c.use_dilation))
        return boxes

    # ---- Classifier::Run (fixed 3x48x192)
    def cls_run(self, crops):
        n = len(crops)
        labels = np.zeros(n, np.int32)
        scores = np.zeros(n, np.float32)
        probs = np.zeros((n, 2), np.float32)
        for beg in range(0, n, self.cls_batch_num):
            end = min(n, beg + self.cls_batch_num)
            x = np.stack([O.cls_preprocess(crops[i]) for i in range(beg, end)])
            p = self.cls.run(x).reshape(end - beg, 2)
            lg = self.cls.logits().reshape(end - beg, 2)
            for j in range(end - beg):
                labels[beg + j] = int(np.argmax(lg[j]))  # first maximum (of the logits), like std::max_element
                scores[beg + j] = p[j][labels[beg + j]]
                probs[beg + j] = p[j]
        self.taps["cls_probs"] = probs
        return labels, scores

    # ---- CRNNRecognizer::Run
    def rec_run(self, crops):
        n = len(crops)
        texts = [np.zeros(0, np.int32) for _ in range(n)]
        scores = np.zeros(n, np.float32)
        steps = [None] * n
        if n == 0:
            return texts, scores, steps
        width_list = np.array([np.float32(c.shape[1]) / np.float32(c.shape[0]) for c in crops], np.float32)
        indices = O.argsort(width_list, stable=self.rec_sort == "stable")
        imgH, imgW = self.rec_img_h, self.rec_img_w
        for beg in range(0, n, self.rec_batch_num):
            end = min(n, beg + self.rec_batch_num)
            max_wh_ratio = np.float32(imgW * 1.0 / imgH)
            for ino in range(beg, end):
                hh, ww = crops[indices[ino]].shape[:2]
                if not self.srv_rec:
                    max_wh_ratio = max(max_wh_ratio, np.float32(ww * 1.0 / hh))
            bw = int(np.float32(imgH) * np.float32(max_wh_ratio))
            tensor_w = max(bw, imgW)
            batch = []
            for ino in range(beg, end):
                xi = O.rec_preprocess(crops[indices[ino]], imgH, bw)
                if tensor_w > bw:  # only when int(imgH*ratio) rounds below rec_img_w; see DESIGN.md quirks
                    pad = np.empty((imgH, tensor_w, 3), np.float32)
                    pad[:] = np.float32(-1.0)
                    pad[:, :bw] = xi
                    xi = pad
                batch.append(xi)
            p = self.rec.run(np.stack(batch))  # [N,1,T,6625]
            if self.srv_rec:
                lg = p
                e = np.exp(lg.astype(np.float64) - lg.max(axis=-1, keepdims=True))
                p = (e / e.sum(axis=-1, keepdims=True)).astype(np.float32)
            else:
                lg = self.rec.logits()
            for m in range(end - beg):
                pm = p[m, 0]
                # arg max = first maximum of the logits (exp is not strictly monotonic after rounding);
                # its probability is the softmax value at that index
                amax = lg[m, 0].argmax(axis=1).astype(np.int32)
                pmax = pm[np.arange(pm.shape[0]), amax].astype(np.float32)
                ids, sc = O.ctc_decode(amax, pmax)
                li = indices[beg + m]
                steps[li] = (amax, pmax)
                if ids is None:
                    continue
                texts[li] = ids
                scores[li] = sc
        return texts, scores, steps

    # ---- OCRWorker::processRequest
    def process(self, image, prob_override=None):
        if image is None or image.size == 0:
            return dict(success=False, error="Empty image data provided")
        image = image.copy()  # OCRRequest clones the Mat (ocr_worker.h:28-29)
        rows, cols = image.shape[:2]
        boxes = self.det_run(image, prob_override)
        words = []
        if len(boxes) == 0:
            return dict(success=True, width=cols, height=rows, words=words)
        views, owner = [], []
        for bi, b in enumerate(boxes):
            if self.crop_mode == "rotate":
                c = O.rotate_crop(image, b)          # utility.cpp:137-190: an independent image per box
                if c is not None:
                    views.append(c)
                    owner.append(bi)
                continue
            r = O.crop_rect(b, rows, cols)
            if r is not None:
                x, y, w, h = r
                owner.append(len(views))             # boxes[i] for text i: reference quirk kept
                views.append(image[y:y + h, x:x + w])
        if not views:
            return dict(success=True, width=cols, height=rows, words=words)
        if self.enable_cls:
            labels, _ = self.cls_run(views)
            for i, v in enumerate(views):
                if labels[i] == 1:
                    O.rotate180_inplace(v)  # in place on the shared image, like cv::rotate on an ROI view
        texts, scores, _ = self.rec_run(views)
        for i in range(len(texts)):
            words.append(dict(ids=texts[i], confidence=float(scores[i]), box=boxes[owner[i]]))
        return dict(success=True, width=cols, height=rows, words=words)
