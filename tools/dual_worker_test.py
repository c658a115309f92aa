"""Development measurement: one pipeline handle over 64 images vs W handles (one host thread each) over 64/W
images each, same GPU.  python tools/dual_worker_test.py [workers]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    pkg = bench.load_package()
    imgs, probs = bench.make_inputs(bench.shard_seeds(0, 1))
    B, H, Wd = bench.BATCH, bench.H, bench.W
    per = B // W
    pipes = [pkg.Pipe(device=0, enable_cls=True, limit_side_len=960, rec_batch_num=16, rec_img_h=48, rec_img_w=320) for _ in range(W)]
    di = [pkg.DevArray(imgs[k * per:(k + 1) * per]) for k in range(W)]
    dp = [pkg.DevArray(probs[k * per:(k + 1) * per]) for k in range(W)]
    words = [0] * W

    def work(k, steps):
        for _ in range(steps):
            words[k] = pipes[k].run_device(di[k], H, Wd, per, dp[k], collect=False)

    def run(steps):
        th = [threading.Thread(target=work, args=(k, steps)) for k in range(W)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        pkg.check(pkg.lib().ocr_dev_sync())
        return time.perf_counter() - t0

    run(2)
    el = run(6)
    print("workers %d x %d images: %.2f ms per %d images, %.1f img/s, words %s" % (W, per, el / 6 * 1e3, B, B * 6 / el, words))


if __name__ == "__main__":
    main()
