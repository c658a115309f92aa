"""One CPU worker of bench.py's cpu_baseline leg (TEST/BENCH INFRASTRUCTURE: uses oracle/, never the product path).

The reference's CPU deployment is N OCRWorker threads, each with its own three predictors and 2 / 1 / 2 math threads for
det / cls / rec (/root/reference/src/ocr_worker.cpp:16-18), N = floor(0.8 * cores / 3) for throughput (:345-349).  Paddle
Inference + MKLDNN and OpenCV cannot run on the box (absent from the reference checkout), so two stand-ins are timed,
both driven by the oracle's restatement of DBDetector / Classifier / CRNNRecognizer::Run and processRequest:

  --engine oracle   B1: the C oracle's plan executor (bit-exact contract arithmetic, OpenMP) - a checker, not a tuned
                    library: a LOWER bound on what the reference's CPU path does
  --engine torch    B2: torch-CPU (oneDNN kernels) interpreting the same .pdmodel graphs - the closest available
                    proxy for Paddle+MKLDNN kernel quality

    python tools/cpu_baseline_worker.py --engine oracle --workload cfg2 --threads 2 --budget 10 --first 0
prints one JSON line {"images": n, "seconds": s, "ms": [...]} when its time budget is spent (at least one image).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")]


class TorchNet:
    """OracleNet's interface (run on NHWC, logits) over graph_ref.run_graph."""

    def __init__(self, kind, weights):
        self.kind = kind
        self.path = os.path.join(ROOT, "models", kind, "inference.pdmodel")
        self.weights = weights
        self._p = None

    def run(self, x_nhwc):
        from graph_ref import run_graph
        y = run_graph(self.path, self.weights, np.ascontiguousarray(x_nhwc.transpose(0, 3, 1, 2)))
        self._p = y
        if self.kind == "det":
            return y.transpose(0, 2, 3, 1)
        if self.kind == "rec":
            return y.reshape(y.shape[0], 1, y.shape[1], y.shape[2])
        return y

    def logits(self):
        # the softmax is monotonic: arg max of log p is the arg max of the logits (timing stand-in: ties do not matter)
        y = np.log(np.maximum(self._p, 1e-38))
        return y.reshape(y.shape[0], 1, y.shape[1], y.shape[2]) if self.kind == "rec" else y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--engine", choices=["oracle", "torch"], default="oracle")
    ap.add_argument("--workload", choices=["cfg1", "cfg2"], default="cfg2")
    ap.add_argument("--threads", type=int, default=2)
    ap.add_argument("--budget", type=float, default=10.0)
    ap.add_argument("--first", type=int, default=0, help="first cfg2 sample index of this worker")
    args = ap.parse_args()
    os.environ["OMP_NUM_THREADS"] = str(args.threads)
    import oracle as O
    from pipeline import Pipeline, DetCfg
    set_threads = lambda n: None
    if args.engine == "torch":
        import torch
        torch.set_num_threads(args.threads)
        set_threads = lambda n: torch.set_num_threads(n)
    else:
        try:
            gomp = ctypes.CDLL("libgomp.so.1")
            set_threads = lambda n: gomp.omp_set_num_threads(int(n))
        except OSError:
            pass
    if args.workload == "cfg1":
        pipe = Pipeline()                                         # the worker's literals, cls off (ocr_worker.cpp:21-63)
    else:
        pipe = Pipeline(det_cfg=DetCfg(limit_side_len=960), rec_batch_num=16, rec_img_h=48, rec_img_w=320, enable_cls=True)
    if args.engine == "torch":
        for k in ("det", "rec", "cls"):
            net = getattr(pipe, k)
            if net is not None:
                setattr(pipe, k, TorchNet(k, net.weights))
    det_t, cls_t, rec_t = max(1, args.threads), max(1, args.threads // 2), max(1, args.threads)   # 2 / 1 / 2
    card = np.load(os.path.join(ROOT, "tests", "golden", "card_jd_bgr.npy"))

    def one(i):
        if args.workload == "cfg1":
            set_threads(det_t)
            pipe.process(card)
            return
        from bench import _cfg2_cached, H, W
        img, prob = _cfg2_cached(args.first + i)
        img = img.copy()
        set_threads(det_t)
        x, _ = O.det_preprocess(img, H, W)
        pipe.det.run(x[None])  # full det network (timed, result replaced by the synthetic map as on the GPU)
        boxes = O.det_post(prob, 0.2, 0.4, 1.8, H, W)
        views = []
        for b in boxes:
            r = O.crop_rect(b, H, W)
            if r:
                xx, yy, ww, hh = r
                views.append(img[yy:yy + hh, xx:xx + ww])
        if views:
            set_threads(cls_t)
            labels, _ = pipe.cls_run(views)
            for k, v in enumerate(views):
                if labels[k] == 1:
                    O.rotate180_inplace(v)
            set_threads(rec_t)
            pipe.rec_run(views)

    one(0)  # warm-up (page-in, oneDNN primitive caches): not counted
    ms = []
    t0 = time.time()
    i = 0
    while True:
        s0 = time.time()
        one(i + 1)
        ms.append((time.time() - s0) * 1e3)
        i += 1
        if time.time() - t0 >= args.budget:
            break
    print(json.dumps({"images": i, "seconds": time.time() - t0, "ms": ms}), flush=True)


if __name__ == "__main__":
    main()
