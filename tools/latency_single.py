"""Single-request latency (one image per call, worker defaults): the card image and one 960x960 synthetic image."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402


def main():
    pkg = bench.load_package()
    from synth_data import cfg2_sample
    card = np.load(os.path.join(ROOT, "tests", "golden", "card_jd_bgr.npy"))
    big = cfg2_sample(0)[0]
    for name, img, kw in (("card-jd 178x391, worker defaults (limit 512, rec 28x192)", card, {}),
                          ("synthetic 960x960, limit 960, rec 48x320, cls on", big,
                           dict(enable_cls=True, limit_side_len=960, rec_img_h=48, rec_img_w=320))):
        pipe = pkg.Pipe(**kw)
        for _ in range(3):
            pipe.run([img])
        ts = []
        for _ in range(20):
            t0 = time.perf_counter()
            w = pipe.run([img])
            ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        print("%-60s p50 %.2f ms  p90 %.2f ms  words %d" % (name, ts[len(ts) // 2], ts[int(len(ts) * 0.9)], len(w[0])))
        pipe.close()


if __name__ == "__main__":
    main()
