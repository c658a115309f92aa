"""Development measurement: wall time and stage times of repeated single-image detector runs on one mixed-size
(configs[2]) image - the latency-bound per-size pass that the detector lanes of ocr_pipe overlap.  Run from the repo root."""
import sys, time, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import numpy as np
from __graft_entry__ import load_package
pkg = load_package()
from synth_data import cfg3_sample
img = cfg3_sample(3)[0]
print(img.shape)
det = pkg.Det(limit_side_len=960)
for it in range(6):
    t0 = time.perf_counter(); b = det.run(img); dt = (time.perf_counter() - t0) * 1e3
    print("run %d wall %.2f ms  pre/infer/post %s  boxes %d" % (it, dt, [round(x, 3) for x in det.times], len(b)))
