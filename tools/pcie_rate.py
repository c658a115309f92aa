"""PCIe-inclusive rate of the bench workload: the 64 host images are handed to ocr_pipe_run (upload inside the
call), no probability-map override.  Reported in DESIGN.md; never `value` of bench.py."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    pkg = bench.load_package()
    imgs, probs = bench.make_inputs(bench.shard_seeds(0, 1))
    pipe = pkg.Pipe(device=0, enable_cls=True, limit_side_len=960, rec_batch_num=16, rec_img_h=48, rec_img_w=320)
    lst = [imgs[i] for i in range(imgs.shape[0])]
    d_i, d_p = pkg.DevArray(imgs), pkg.DevArray(probs)
    t0 = time.perf_counter()
    for _ in range(4):
        pkg.check(pkg.lib().ocr_dev_upload(d_i.ptr, imgs.ctypes.data, imgs.nbytes))
    up = (time.perf_counter() - t0) / 4
    print("H2D of the 64 images (%.0f MB, pageable host memory): %.2f ms = %.1f GB/s" % (imgs.nbytes / 1e6, up * 1e3, imgs.nbytes / up / 1e9))
    for mode in ("device-resident + map override (bench protocol)", "host images, upload inside the call, network's own map"):
        f = (lambda: pipe.run_device(d_i, bench.H, bench.W, bench.BATCH, d_p, collect=False)) if mode.startswith("device") else \
            (lambda: sum(len(w) for w in pipe.run(lst)))
        f()
        t0 = time.perf_counter()
        n = 4
        for _ in range(n):
            words = f()
        dt = (time.perf_counter() - t0) / n
        print("%-60s %.2f ms/step  %.1f img/s  words %d" % (mode, dt * 1e3, bench.BATCH / dt, words))


if __name__ == "__main__":
    main()
