"""Development probe: aggregate cfg2 throughput of W pipeline workers sharing ONE GPU (the reference lets several
workers share a device: worker i -> GPU i mod n, gpu_worker_pool.cpp:46-59), each with its own ocr_pipe and streams.

    python tools/inflight_probe.py --workers 2 --steps 6
"""
import argparse
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, default=2)
    ap.add_argument("--steps", type=int, default=6)
    args = ap.parse_args()
    import bench
    import importlib
    pkg = importlib.import_module("cpp-paddle-ocr_amd")
    imgs, probs = bench.make_inputs(list(range(bench.BATCH)), workers=8)
    img_list, prob_list = [imgs[i] for i in range(bench.BATCH)], [probs[i] for i in range(bench.BATCH)]
    for W in sorted({1, args.workers}):
        pipes = [pkg.Pipe(device=0, enable_cls=True, limit_side_len=960, rec_batch_num=16, rec_img_h=48, rec_img_w=320) for _ in range(W)]
        for p in pipes:
            p.stage(0, img_list, prob_list)
            p.run_staged(0, collect=False)
            p.run_staged(0, collect=False)
        pkg.dev_sync() if hasattr(pkg, "dev_sync") else None
        def work(p, n):
            for _ in range(n):
                p.run_staged(0, collect=False)
        ths = [threading.Thread(target=work, args=(p, args.steps)) for p in pipes]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        el = time.perf_counter() - t0
        print("workers %d: %d steps each, %.1f ms per step (aggregate), %.1f images/s" % (
            W, args.steps, el * 1e3 / (W * args.steps), bench.BATCH * W * args.steps / el), flush=True)
        for p in pipes:
            p.close()


if __name__ == "__main__":
    main()
